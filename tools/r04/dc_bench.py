#!/usr/bin/env python3
"""Host-side pieces of a whole-file decode on ONE thread, no GPU needed: every DC group of tests/data/e2e_8k_d1.jxl
(jxlhip_dc_group_decode), the AC-global section and every AC group (jxlhip_ac_group_decode_sparse).  With
JXLHIP_LIB=<path> a library built with -DJXLHIP_DC_TIMING prints the per-channel milliseconds of the DC groups.
usage: tools/r04/dc_bench.py [reps]"""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from libjxl_amd import abi
from test_dc_groups import parse_to_sections

if os.environ.get("JXLHIP_LIB"):
    abi._SO = os.environ["JXLHIP_LIB"]
L = abi.load_library()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lenient = "--dc-only-lenient" in sys.argv  # (timing experiments that decode garbage: DC groups only, any status)


class _File:
    codestream = np.fromfile(os.path.join(ROOT, "tests", "data", "e2e_8k_d1.jxl"), np.uint8)


cs, ih, fh, sections = parse_to_sections(L, _File())
s0 = sections[0]
dcg, dpos = abi.DcGlobal(), C.c_size_t(0)
assert L.jxlhip_dc_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), fh.flags, C.byref(dcg)) == 0
tree = C.c_void_p()
assert L.jxlhip_modular_global_decode(s0.ctypes.data, len(s0), C.byref(dpos), C.byref(fh), C.byref(tree)) == 0
xsb, ysb = fh.xsize_blocks, fh.ysize_blocks
qdc = [np.zeros(xsb * ysb, np.int32) for _ in range(3)]
acs, rq, sharp = np.zeros(xsb * ysb, np.uint8), np.zeros(xsb * ysb, np.int32), np.zeros(xsb * ysb, np.uint8)
cw, chh = (xsb + 7) // 8, (ysb + 7) // 8
ytox, ytob = np.zeros(cw * chh, np.int8), np.zeros(cw * chh, np.int8)
used = C.c_uint32(0)
ndc, ng, npass = int(fh.num_dc_groups), int(fh.num_groups), int(fh.num_passes)
best = None
for rep in range(reps):
    ts = []
    for g in range(ndc):
        d = sections[1 + g]
        gp, ep = C.c_size_t(0), C.c_uint32(0)
        ptrs = (C.c_void_p * 3)(*[q.ctypes.data for q in qdc])
        t0 = time.perf_counter()
        rc = L.jxlhip_dc_group_decode(tree, d.ctypes.data, len(d), C.byref(gp), C.byref(fh), g, ptrs, C.byref(ep), acs.ctypes.data,
                                      rq.ctypes.data, sharp.ctypes.data, ytox.ctypes.data, ytob.ctypes.data, C.byref(used))
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0 or lenient
    best = ts if best is None else [min(a, b) for a, b in zip(best, ts)]
print("DC groups, best of %d, ms each: %s; sum %.1f" % (reps, " ".join("%.2f" % t for t in best), sum(best)))
h = hashlib.sha1()
for a in qdc + [acs, rq, sharp, ytox, ytob]:
    h.update(a.tobytes())
print("digest of the side info", h.hexdigest())
if lenient:
    sys.exit(0)

sg = sections[1 + ndc]
enc = (abi.QuantEncoding * 17)()
ts = []
for rep in range(reps):
    passes = (C.c_void_p * 11)()
    nh, bits = C.c_uint32(0), C.c_size_t(0)
    t0 = time.perf_counter()
    assert L.jxlhip_ac_global_decode(sg.ctypes.data, len(sg), ng, npass, used.value, C.byref(dcg.block_ctx_map), enc, C.byref(nh), passes,
                                     C.byref(bits)) == 0
    ts.append((time.perf_counter() - t0) * 1e3)
    if rep + 1 < reps:
        for q in passes:
            if q:
                L.jxlhip_ac_pass_destroy(q)
print("AC global: %.2f ms (best of %d)" % (min(ts), reps))
qctx = np.zeros(xsb * ysb, np.uint8)
q3 = (C.c_void_p * 3)(*[q.ctypes.data for q in qdc])
assert L.jxlhip_quant_dc_contexts(C.byref(dcg.block_ctx_map), xsb * ysb, q3, qctx.ctypes.data) == 0
xsg = (fh.xsize + 255) // 256
caps = (C.c_uint32 * 3)(16382, 65536, 16382)
ents = [np.zeros(c, np.uint32) for c in caps]
entp = (C.c_void_p * 3)(*[e.ctypes.data for e in ents])
for rep in range(min(reps, 3)):
    t0 = time.perf_counter()
    worst, nz, dense = 0.0, 0, 0
    for g in range(ng):
        d = sections[2 + ndc + g]
        cnt, pos, nco = (C.c_uint32 * 3)(), C.c_size_t(0), C.c_size_t(0)
        t1 = time.perf_counter()
        rc = L.jxlhip_ac_group_decode_sparse(passes[0], xsb, ysb, g % xsg, g // xsg, acs.ctypes.data, rq.ctypes.data, qctx.ctypes.data,
                                             d.ctypes.data, len(d), C.byref(pos), 0, entp, caps, cnt, C.byref(nco))
        worst = max(worst, time.perf_counter() - t1)
        if rc == -8:
            dense += 1
            continue
        assert rc == 0, (g, rc)
        nz += sum(cnt)
    dt = time.perf_counter() - t0
    print("AC groups: %d in %.1f ms, %.3f ms each (worst %.3f), %d too dense for the sparse form, %.1f ns per byte" % (
        ng, dt * 1e3, dt * 1e3 / ng, worst * 1e3, dense, dt * 1e9 / sum(len(sections[2 + ndc + g]) for g in range(ng))))
