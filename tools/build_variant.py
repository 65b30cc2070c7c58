#!/usr/bin/env python3
"""Experiment builds: tools/build_variant.py <tag> [--only a.hip,b.hip] <extra hipcc flags...>
-> libjxl_amd/csrc/variants/libjxl_hip_<tag>.so (select at run time with JXLHIP_SO=<path>).
Only kernels_*.hip see the extra flags; with --only, just the named sources are recompiled and every other
object is the product build's (libjxl_amd/csrc/build/*.o, which must be up to date)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libjxl_amd import build as B
tag, extra = sys.argv[1], sys.argv[2:]
only = None
if extra and extra[0] == "--only":
    only, extra = set(extra[1].split(",")), extra[2:]
    B.build()
out = os.path.join(B.CSRC, "variants"); os.makedirs(out, exist_ok=True)
bdir = os.path.join(out, "build_" + tag); os.makedirs(bdir, exist_ok=True)
def comp(src):
    if only is not None and src not in only:
        return os.path.join(B.BUILD, os.path.splitext(src)[0] + ".o")
    obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
    lang = ["-x", "hip"] if src.endswith(".hip") else []
    fl = B.FLAGS + (extra if src.startswith("kernels_") else [])
    r = subprocess.run([B.HIPCC] + fl + lang + ["-c", os.path.join(B.CSRC, src), "-o", obj], capture_output=True, text=True)
    if r.returncode: raise RuntimeError(r.stderr[-3000:])
    return obj
with ThreadPoolExecutor(6) as ex: objs = list(ex.map(comp, B.LIB_SOURCES))
so = os.path.join(out, f"libjxl_hip_{tag}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs, check=True)
print(so)
