#!/usr/bin/env python3
"""Root cause of the order-dependent reference result, proven by a one-condition patch.

LowMemoryRenderPipeline::RenderRect lets a stage run `xextra_right` columns past its rect
(low_memory_render_pipeline.cc:751-757) but mirrors the stage's input at the right image edge only when
rect.x1 + border_x >= image_xsize (ApplyXMirroring, :496/:510) -- xextra_right is not part of the test.  A rect
that ends within (xextra_right + border) of the edge but not within border of it makes the stage read one or more
columns past the image edge unmirrored: stale bytes of the thread's stage buffer.  With groups finishing in index
order no such rect exists; it appears when the narrow last group column (width < 16 + total filter border)
finishes BEFORE its left neighbour, whose 32-px border strip [x1-16, x1+16) is then rendered on its own.

This script builds oracle/_ref/libjxl_ref_mirrorfix.so from the same objects with a patched COPY of that one
translation unit (the copy goes to oracle/_build/, never into the repo) and shows that the order dependence
disappears.  usage: ref_mirror_fix_probe.py"""
import itertools, os, re, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref

src = os.path.join(build_ref.REF, "lib/jxl/render_pipeline/low_memory_render_pipeline.cc")
text = open(src).read()
# mirror whenever the stage may READ past the edge: rect.x1 + xextra (<= the padding the later stages need) + border
old = "ApplyXMirroring(input_rows[i][c][iy], stages_[i]->settings_.border_x,"
new = "ApplyXMirroring(input_rows[i][c][iy], stages_[i]->settings_.border_x + xpadding_for_output_[i],"
assert text.count(old) == 1
bdir = os.path.join(ROOT, "oracle", "_build", "mirrorfix"); os.makedirs(bdir, exist_ok=True)
patched = os.path.join(bdir, "low_memory_render_pipeline_mirrorfix.cc")
open(patched, "w").write(text.replace(old, new))
obj = os.path.join(bdir, "lmrp_fix.o")
subprocess.check_call([build_ref.CXX] + build_ref.FLAGS + ["-c", patched, "-o", obj])
objs = [o for o in build_ref.build(only_compile=True) if "low_memory_render_pipeline" not in o]
objs += [os.path.join(build_ref.OBJ, d + ".o") for d in ("ref_driver", "ref_real_stream")] + [obj]
lib = os.path.join(build_ref.OUT, "libjxl_ref_mirrorfix.so")
subprocess.check_call([build_ref.CXX, "-shared", "-fPIC", "-o", lib] + objs + ["-Wl,--gc-sections", "-Wl,--no-undefined", "-lpthread", "-lm"])

import ctypes as C
import oracle, frames
from libjxl_amd import synth
oracle.ref_threads = lambda x, y, t: t
stock = oracle.ref_lib()
fixed = C.CDLL(lib)
fixed.jxr_decode_frame.argtypes = stock.jxr_decode_frame.argtypes
def set_ref_lib(L):
    oracle._ref = L
oracle.set_ref_lib = set_ref_lib
def run(fr, L, order=None, threads=1):
    if order: os.environ["JXR_GROUP_ORDER"] = order
    else: os.environ.pop("JXR_GROUP_ORDER", None)
    return fr.decode_ref(threads=threads)
for size, gab, epf in [((533, 401), 1, 3), ((530, 300), 1, 1), ((530, 300), 1, 2), ((530, 300), 1, 3)]:
    params, t, fr = frames.make_case(*size, mix=synth.MIX_ALL, gab=bool(gab), epf_iters=epf, seed=21 + gab + 2 * epf)
    c = fr.decode(threads=4)
    for name, L in (("stock", stock), ("mirrorfix", fixed)):
        oracle.set_ref_lib(L)
        a = run(fr, L)
        b = run(fr, L, "0,2,1")
        m = run(fr, L, None, threads=8)
        print("%s %s gab=%d epf=%d: in-order vs C %.3g | order 0,2,1 vs C %.3g | 8 threads vs C %.3g" % (
            name, size, gab, epf, np.abs(a - c).max(), np.abs(b - c).max(), np.abs(m - c).max()))
oracle.set_ref_lib(stock)
