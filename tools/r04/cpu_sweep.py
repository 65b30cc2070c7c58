#!/usr/bin/env python3
"""GPU box (its host CPU): the reference's decode (oracle/_ref) by build (one lane -O3 -mavx2 -mfma / 8-lane hot path) and
thread count, on the 4096x2160 sample bench.py used so far and on the whole 8K frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import frames, oracle
from libjxl_amd import synth
print("cpus", os.cpu_count(), flush=True)
for (w, h) in ((4096, 2160), (7680, 4320)):
    _, _, fr = frames.make_case(w, h, mix=synth.MIX_D1, gab=True, epf_iters=1)
    for name, kw in (("fma", dict(fma_build=True)), ("v8", dict(v8_build=True))):
        row = []
        for thr in (1, 8, 16, 32, 64, 128, 256):
            if thr > (os.cpu_count() or 1):
                continue
            fr.decode_ref(threads=thr, **kw)
            n = 1 if thr == 1 else 4
            t0 = time.perf_counter()
            for _ in range(n):
                fr.decode_ref(threads=thr, **kw)
            row.append("%d: %.0f" % (thr, w * h * n / (time.perf_counter() - t0) / 1e6))
        print("%dx%d %s  Mpx/s by threads  %s" % (w, h, name, "  ".join(row)), flush=True)
