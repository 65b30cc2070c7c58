"""GPU box: achievable HBM bandwidth of plain torch kernels (copy / fill / read-reduce), to calibrate rooflines."""
import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
for mb in (400, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_(); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); tf = t(lambda: b.fill_(1.0)); tr = t(lambda: a.sum())
    print(f"{mb} MiB: copy {2*n*4/tc/1e12:.2f} TB/s (r+w)  fill {n*4/tf/1e12:.2f} TB/s  sum {n*4/tr/1e12:.2f} TB/s")
