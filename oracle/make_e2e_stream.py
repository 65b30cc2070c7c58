"""Writes the whole-file workload of bench.py's `e2e` block and of the 8K tests: a genuine 7680x4320 RGB VarDCT
codestream (d1.0, effort 7 = libjxl's default) made by the REFERENCE ENCODER (oracle/_ref/libjxl_ref.so =
/root/reference/lib/jxl compiled in place, oracle/build_ref.py) from a procedural image -> tests/data/e2e_8k_d1.jxl
(the file is committed: a fixture like tests/data/real_4k_d1.npz, with this script as the record of how it was made).

Test / measurement infrastructure, not product code.  Runs wherever oracle/_ref/libjxl_ref.so exists (the build
container; the GPU box gets the prebuilt library AND the finished file with the snapshot, and can re-make the file with
this script if it is missing: ~3 minutes of one core).
usage: python oracle/make_e2e_stream.py [out.jxl [xsize ysize distance speed_tier]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEFAULT = os.path.join(ROOT, "tests", "data", "e2e_8k_d1.jxl")  # committed (3.1 MB): the GPU box needs no encoder run


def make(out=DEFAULT, xsize=7680, ysize=4320, distance=1.0, speed_tier=3, seed=7):
    import oracle
    oracle.ref_lib()
    rs = oracle.RealStream(xsize, ysize, seed=seed, distance=distance, speed_tier=speed_tier)
    tmp = out + ".tmp"
    with open(tmp, "wb") as f:
        f.write(rs.codestream.tobytes())
    os.replace(tmp, out)
    return out, rs


def ensure(out=DEFAULT, generate=True):
    """The file's path, made on first use (about three minutes of one core: only with generate=True); None when it is
    absent and cannot / may not be made."""
    if os.path.exists(out):
        return out
    import oracle
    if not generate or not oracle.ref_available():
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
