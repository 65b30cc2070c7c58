import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle as O
from libjxl_amd import abi
L = abi.load_library()
lum = (C.c_float * 3)(0.2126, 0.7152, 0.0722)
def run(kw, own_stream, label, first_f32=False):
    rs = O.RealStream(**kw)
    cs = rs.codestream.tobytes()
    W, H = kw["xsize"], kw["ysize"]
    ctx = C.c_void_p()
    assert L.jxlhip_create(0, C.byref(ctx)) == 0
    if not own_stream:
        L.jxlhip_set_stream(ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream), 0)
    torch.cuda.synchronize()
    if first_f32:
        f = abi.OutputFormat(1, 0, 4, 32, 0, 0.0, lum)
        o = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
        assert L.jxlhip_decode_codestream(ctx, None, None, cs, len(cs), 2, C.byref(f), o.data_ptr(), W * 16, 0, None) == 0
    fmt8 = abi.OutputFormat(1, 1, 4, 8, 0, 0.0, lum)
    out8 = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    rc = L.jxlhip_decode_codestream(ctx, None, None, cs, len(cs), 2, C.byref(fmt8), out8.data_ptr(), W * 4, 0, None)
    torch.cuda.synchronize()
    g8 = out8.cpu().numpy()
    want = np.clip(rs.rgb, 0, 1) * 255.0
    d = np.abs(g8[..., :3].astype(np.float32) - want)
    print("%-40s rc %d rgb max diff %.1f bad frac %.3f alpha ok %s" % (label, rc, d.max(), (d.max(axis=2) > 1.6).mean(),
          np.array_equal(g8[..., 3], np.rint(rs.alpha * 255).astype(np.uint8)) if rs.alpha is not None else (g8[..., 3] == 255).all()), flush=True)
    L.jxlhip_destroy(ctx)
base = dict(seed=12, xsize=200, ysize=120, distance=1.0, speed_tier=3, original="srgb8")
run(dict(base, alpha_bits=8), True, "alpha, own stream, u8 first")
run(dict(base, alpha_bits=8), False, "alpha, torch stream, u8 first")
run(dict(base), True, "no alpha, own stream, u8 first")
run(dict(base, alpha_bits=8), True, "alpha, own stream, f32 then u8", first_f32=True)
run(dict(base, alpha_bits=8, xsize=520, ysize=300), True, "alpha 520x300, own stream, u8 first")
os.environ["JXLHIP_FILTERS"] = "generic"
run(dict(base, alpha_bits=8), True, "alpha, own stream, generic filters")
