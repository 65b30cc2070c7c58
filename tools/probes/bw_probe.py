import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
for mb in (100, 400, 1600):
    n = mb*1024*1024//4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    print(f"{mb} MB: fill {mb/1024/tf:.0f} GB/s  copy(r+w) {2*mb/1024/tc:.0f} GB/s  read-sum {mb/1024/tr:.0f} GB/s")
