#!/usr/bin/env python3
"""GPU box: does data written by one kernel come back faster when the next kernel reads it soon enough?  Copies
z -> x then x -> y for growing buffer sizes: below the L2s (8 x 4 MB) and below the 256 MB memory-side cache the second
copy's reads could be served without HBM.  Prints the rate of the pair (4 x size bytes moved) per size."""
import time
import torch

dev = "cuda:0"
for mb in (8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    z = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    x = torch.empty_like(z)
    y = torch.empty_like(z)
    reps = max(10, 4096 // mb)
    for _ in range(3):
        x.copy_(z); y.copy_(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        x.copy_(z)
        y.copy_(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    # the same bytes with no producer -> consumer reuse: two independent copies out of buffers last touched 2 copies ago
    t0 = time.perf_counter()
    for _ in range(reps):
        x.copy_(z)
    torch.cuda.synchronize()
    d1 = (time.perf_counter() - t0) / reps
    print("%5d MB: write-then-read pair %.2f TB/s   plain copy %.2f TB/s" % (mb, 4 * n * 4 / dt / 1e12, 2 * n * 4 / d1 / 1e12), flush=True)
    del x, y, z
