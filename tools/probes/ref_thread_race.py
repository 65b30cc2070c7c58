#!/usr/bin/env python3
"""Reproducer for the threaded-reference NaN column: oracle/_ref with T threads against the
same library with 1 thread, same frame, repeated.  usage: ref_thread_race.py [threads] [repeats] [xsize ysize]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import oracle, frames
from libjxl_amd import synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
xs, ys = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (533, 401)
oracle.ref_threads = lambda x, y, t: t   # the workaround under test: off
bad = 0
for gab, epf in [(1, 1), (1, 3), (0, 2), (1, 0), (0, 0)]:
    params, t, fr = frames.make_case(xs, ys, mix=synth.MIX_ALL, gab=bool(gab), epf_iters=epf, seed=21 + gab + 2 * epf)
    one = fr.decode_ref(threads=1)
    for i in range(N):
        many = fr.decode_ref(threads=T)
        if not np.array_equal(one, many, equal_nan=True):
            d = np.argwhere(~np.isclose(one, many, rtol=0, atol=0, equal_nan=True))
            print("MISMATCH gab=%d epf=%d run %d: %d samples, columns %s rows %d..%d nan=%d" % (
                gab, epf, i, len(d), sorted(set(d[:, 1].tolist()))[:8], d[:, 0].min(), d[:, 0].max(), int(np.isnan(many).sum())))
            bad += 1
    print("gab=%d epf=%d: %d runs done, mismatching so far %d" % (gab, epf, N, bad), flush=True)
print("total mismatching runs:", bad)
