#!/usr/bin/env python3
"""GPU box: c3 frames decoded with 1, 2, 3 frames in flight (one VarDctDecoder + stream per slot, shared read-only inputs).
Launch gaps, k_prepare and kernel tails of frame k overlap the kernels of frame k+1 -- what a server decoding a queue of
images sees.  usage: tools/r04/inflight.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from libjxl_amd import VarDctDecoder, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
xs, ys = 7680, 4320
params, t = synth.synth_frame(xs, ys, mix=synth.MIX_D1, gab=True, epf_iters=1, device="cuda:0")
for slots in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(slots)]
    decs, outs = [], []
    for s in streams:
        with torch.cuda.stream(s):
            d = VarDctDecoder(0)
        d.begin_frame(params)
        d.set_inputs(t, d.default_dequant_tables() if not decs else dq)
        if not decs:
            dq = d.default_dequant_tables()
            d.set_inputs(t, dq)
        decs.append(d)
        outs.append(d.alloc_output())
    torch.cuda.synchronize()
    for k in range(20):
        decs[k % slots].decode_frame(outs[k % slots])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        decs[k % slots].decode_frame(outs[k % slots])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("frames in flight %d: %.1f us per frame, %.1f Gpx/s" % (slots, dt * 1e6, xs * ys / dt / 1e9), flush=True)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    for d in decs:
        d.sync()
        d.close()
