#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
nproc; grep -m1 "model name" /proc/cpuinfo
python - <<'PY'
import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import frames, oracle
from libjxl_amd import synth
for (w, h) in ((4096, 2160), (7680, 4320)):
    _, _, fr = frames.make_case(w, h, mix=synth.MIX_D1, gab=True, epf_iters=1)
    for thr in (16, 32, 64, 128, 256):
        fr.decode_ref(threads=thr, fma_build=True)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 3.0:
            fr.decode_ref(threads=thr, fma_build=True); n += 1
        dt = (time.perf_counter() - t0) / n
        print("%dx%d threads %3d: %.1f Mpx/s" % (w, h, thr, w * h / dt / 1e6), flush=True)
PY
