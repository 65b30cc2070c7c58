#!/usr/bin/env python3
"""GPU box: jxlhip_decode_codestream on the 8K stream, N repetitions on one warm runner pool, the per-phase ms of every
repetition; with JXLHIP_CODESTREAM_VERBOSE=1 the library adds the runner timelines of the DC and AC phases (stderr).
usage: tools/r04/e2e_timeline.py [threads] [reps] [gap_ms between repetitions: one cgroup period keeps the CPU quota out of it]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from libjxl_amd import VarDctDecoder, abi

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gap = float(sys.argv[3]) * 1e-3 if len(sys.argv) > 3 else 0.0
blob = open(os.path.join(ROOT, "tests", "data", "e2e_8k_d1.jxl"), "rb").read()
L = abi.load_library()
info = abi.CodestreamInfo()
assert L.jxlhip_codestream_basic_info(blob, len(blob), C.byref(info)) == 0
w, h = info.xsize, info.ysize
R = C.CDLL(abi.runner_library_path())
R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p)
dec = VarDctDecoder(0)
out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
pool = R.JxlThreadParallelRunnerCreate(None, threads)
names = ["headers", "dc_groups", "ac_global", "side_info", "ac_groups", "extra", "kernels_sync"]
tot = []
for rep in range(reps):
    time.sleep(gap)
    t0 = time.perf_counter()
    rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, blob, len(blob), 1, None, out.data_ptr(), w * 12, 0, C.byref(info))
    dt = time.perf_counter() - t0
    assert rc == 0, L.jxlhip_last_error(dec.ctx)
    ms = (C.c_double * 7)()
    L.jxlhip_codestream_phase_ms(dec.ctx, ms)
    print("rep %2d: %.2f ms  " % (rep, dt * 1e3) + " ".join("%s %.2f" % (n, v) for n, v in zip(names, ms) if v >= 0.005), flush=True)
    sys.stderr.flush()
    tot.append(dt * 1e3)
R.JxlThreadParallelRunnerDestroy(pool)
steady = sorted(tot[3:])
print("after 3 warm-up repetitions: median %.2f ms, best %.2f, worst %.2f" % (steady[len(steady) // 2], steady[0], steady[-1]))
