import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle as O
from libjxl_amd import abi, VarDctDecoder
L = abi.load_library()
lum = (C.c_float * 3)(0.2126, 0.7152, 0.0722)
def run(kw, nch, label):
    rs = O.RealStream(**kw)
    cs = rs.codestream.tobytes()
    dec = VarDctDecoder(0)
    W, H = kw["xsize"], kw["ysize"]
    if nch == 4:
        fmt = abi.OutputFormat(0, 0, 4, 32, 0, 0.0, lum)
        out = torch.full((H, W, 4), -7.0, dtype=torch.float32, device="cuda")
        rc = L.jxlhip_decode_codestream(dec.ctx, None, None, cs, len(cs), 2, C.byref(fmt), out.data_ptr(), W * 16, 0, None)
    else:
        out = torch.full((H, W, 3), -7.0, dtype=torch.float32, device="cuda")
        rc = L.jxlhip_decode_codestream(dec.ctx, None, None, cs, len(cs), 1, None, out.data_ptr(), W * 12, 0, None)
    got = out.cpu().numpy()
    d = np.abs(got[..., :3] - rs.rgb).max(axis=2)
    bad = d > 1e-4
    ys, xs = np.nonzero(bad)
    msg = "%s rc %d rgb max diff %.3g bad px %d" % (label, rc, d.max(), bad.sum())
    if bad.any():
        msg += " bbox x %d..%d y %d..%d" % (xs.min(), xs.max(), ys.min(), ys.max())
        gb = np.zeros(((H + 255) // 256, (W + 255) // 256), int)
        for gy in range(gb.shape[0]):
            for gx in range(gb.shape[1]):
                gb[gy, gx] = bad[gy * 256:(gy + 1) * 256, gx * 256:(gx + 1) * 256].sum()
        msg += " per group " + str(gb.tolist())
    if nch == 4 and rs.alpha is not None:
        msg += " alpha max diff %.3g" % float(np.abs(got[..., 3] - rs.alpha).max())
    print(msg, flush=True)
    dec.close()
base = dict(seed=31, xsize=776, ysize=520, distance=2.0, epf=1, speed_tier=3)
for env in ({}, {"JXLHIP_SPARSE_UPLOAD": "0"}):
    os.environ.update(env)
    print("env", env)
    run(dict(base), 4, "no alpha, RGBA out")
    run(dict(base, alpha_bits=16), 3, "alpha16, RGB out")
    run(dict(base, alpha_bits=16), 4, "alpha16, RGBA out")
    run(dict(base, alpha_bits=8), 4, "alpha8, RGBA out")
    run(dict(base, alpha_bits=16, epf=-1), 4, "alpha16 epf auto, RGBA out")
    run(dict(base, alpha_bits=16, distance=1.0), 4, "alpha16 d1, RGBA out")
