"""Writes the whole-file workload of bench.py's `e2e` block and of the 8K tests: a genuine 7680x4320 RGB VarDCT
codestream (d1.0, effort 7 = libjxl's default) made by the REFERENCE ENCODER (oracle/_ref/libjxl_ref.so =
/root/reference/lib/jxl compiled in place, oracle/build_ref.py) from a procedural image -> oracle/_ref/e2e_8k_d1.jxl.

Test / measurement infrastructure, not product code.  Runs wherever oracle/_ref/libjxl_ref.so exists (the build
container; the GPU box gets the prebuilt library AND the finished file with the snapshot, and can re-make the file with
this script if it is missing: ~3 minutes of one core).
usage: python oracle/make_e2e_stream.py [out.jxl [xsize ysize distance speed_tier]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEFAULT = os.path.join(ROOT, "oracle", "_ref", "e2e_8k_d1.jxl")


def make(out=DEFAULT, xsize=7680, ysize=4320, distance=1.0, speed_tier=3, seed=7):
    import oracle
    oracle.ref_lib()
    rs = oracle.RealStream(xsize, ysize, seed=seed, distance=distance, speed_tier=speed_tier)
    tmp = out + ".tmp"
    with open(tmp, "wb") as f:
        f.write(rs.codestream.tobytes())
    os.replace(tmp, out)
    return out, rs


def ensure(out=DEFAULT):
    """The file's path, made on first use; None when the reference encoder is not available."""
    if os.path.exists(out):
        return out
    import oracle
    if not oracle.ref_available():
        return None
    return make(out)[0]


if __name__ == "__main__":
    a = sys.argv[1:]
    out = a[0] if a else DEFAULT
    kw = {}
    if len(a) >= 5:
        kw = dict(xsize=int(a[1]), ysize=int(a[2]), distance=float(a[3]), speed_tier=int(a[4]))
    path, rs = make(out, **kw)
    import numpy as np
    print("wrote", path, os.path.getsize(path), "bytes;", rs.num_groups, "groups; strategies (first cells):",
          np.bincount(rs.ac_strategy.ravel()[(rs.ac_strategy.ravel() & 1) == 1] >> 1, minlength=27).tolist())
