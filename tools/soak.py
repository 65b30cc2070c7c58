"""GPU box: a randomized soak of the whole-file path.  N genuine streams from the reference encoder (random size,
distance, effort, EPF setting, progressive mode -- extra channels then squeezed --, alpha on / off, up to three more
extra channels, sRGB / float / ICC original) through jxlhip_decode_codestream_extra -- ONE context for all of them,
random worker counts, RGB float and RGBA8 outputs, every extra channel as a host plane --
against the pixels the reference decoder produced for the same bytes.  usage: python tools/soak.py [N] [seed]"""
import ctypes as C
import random
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle  # noqa: E402
from libjxl_amd import VarDctDecoder, abi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = abi.load_library()
oracle.ref_lib()
R = C.CDLL(abi.runner_library_path())
R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
pools = {w: R.JxlThreadParallelRunnerCreate(None, w) for w in (3, 8, 24)}
runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p)
lum = (C.c_float * 3)(0.2126, 0.7152, 0.0722)
dec = VarDctDecoder(0)
bad = 0
for i in range(N):
    kw = dict(xsize=rng.choice((97, 200, 256, 257, 520, 777, 1030, 2200)), ysize=rng.choice((64, 120, 256, 300, 513, 776)),
              seed=rng.randrange(1000), distance=rng.choice((0.5, 1.0, 1.0, 2.0, 4.0)), speed_tier=rng.choice((3, 4, 5, 7)),
              epf=rng.choice((-1, -1, 0, 1, 2, 3)), progressive=rng.choice((0, 0, 0, 1, 2)))
    alpha = rng.random() < 0.5
    if alpha:  # (with a progressive mode the extra channels are squeezed)
        kw.update(alpha_bits=rng.choice((8, 16)), alpha_levels=rng.choice((0, 0, 2, 5)))
    more = rng.choice((0, 0, 0, 1, 3))  # extra channels behind the alpha channel: depth / thermal / optional
    if more:
        kw["extra"] = more
    original = rng.choice((None, "srgb8", "srgb16"))
    if original:
        kw["original"] = original
    icc = rng.random() < 0.15  # an ICC original: both decoders (no CMS) write linear sRGB
    if icc:
        sys.path.insert(0, "tests")
        from test_icc import make_profile
        kw["icc"] = make_profile(False, rng.choice((1, 256, 1024)))
    try:
        rs = oracle.RealStream(**kw)
    except ValueError:
        print(i, kw, "encoder refused")
        continue
    cs = rs.codestream.tobytes()
    W, H = kw["xsize"], kw["ysize"]
    w = rng.choice((0, 3, 8, 24))
    args = (runner, pools[w]) if w else (None, None)
    encoded = original is not None and not icc
    scale = max(1.0, float(np.abs(rs.rgb).max()))
    msgs = []
    # float RGB(A) in the space the reference wrote
    nch = 4 if alpha else rng.choice((3, 4))
    fmt = abi.OutputFormat(1 if encoded else 0, 0, nch, 32, 0, 0.0, lum)
    out = torch.full((H, W, nch), -7.0, dtype=torch.float32, device="cuda")
    planes = np.full((4, H, W), -3.0, np.float32)
    ptrs = (C.c_void_p * 4)(*[planes[k].ctypes.data for k in range(4)])
    rc = L.jxlhip_decode_codestream_extra(dec.ctx, *args, cs, len(cs), 2, C.byref(fmt), out.data_ptr(), W * 4 * nch, 0,
                                          ptrs, 4, W, None)
    if not rc:
        want_planes = ([rs.alpha.reshape(H, W)] if alpha else []) + list(rs.extra.reshape(-1, H, W))
        for k, wp in enumerate(want_planes):
            if not np.array_equal(planes[k], wp):
                msgs.append("extra channel %d differs" % k)
    if rc:
        msgs.append("rc %d %s" % (rc, L.jxlhip_last_error(dec.ctx).decode()))
    else:
        got = out.cpu().numpy()
        e = float(np.abs(got[..., :3] - rs.rgb).max()) / scale
        if e > (1e-4 if encoded else 2e-5):
            msgs.append("rgb err %.3g" % e)
        if nch == 4 and not np.array_equal(got[..., 3], rs.alpha if alpha else np.ones((H, W), np.float32)):
            msgs.append("alpha differs")
    # 8-bit sRGB RGBA
    fmt8 = abi.OutputFormat(1, 1, 4, 8, 0, 0.0, lum)
    out8 = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    rc = L.jxlhip_decode_codestream(dec.ctx, *args, cs, len(cs), 2, C.byref(fmt8), out8.data_ptr(), W * 4, 0, None)
    if rc:
        msgs.append("u8 rc %d" % rc)
    else:
        g8 = out8.cpu().numpy()
        lin = np.clip(rs.rgb, 0, 1)
        want = (lin if encoded else np.where(lin <= 0.0031308, lin * 12.92, 1.055 * np.power(lin, 1 / 2.4) - 0.055)) * 255.0
        if np.abs(g8[..., :3].astype(np.float32) - want).max() > 1.6:
            msgs.append("u8 rgb off by %.1f" % np.abs(g8[..., :3].astype(np.float32) - want).max())
        wa = np.rint(rs.alpha * 255.0) if alpha else np.full((H, W), 255.0)
        if np.abs(g8[..., 3].astype(np.float32) - wa).max() > (0 if (not alpha or kw["alpha_bits"] == 8) else 1):
            msgs.append("u8 alpha differs")
    if msgs:
        bad += 1
    if "icc" in kw:
        kw["icc"] = "%d bytes" % len(kw["icc"])
    print(i, "FAIL" if msgs else "ok", kw, "workers", w, msgs, flush=True)
dec.close()
print("soak: %d streams, %d failed" % (N, bad))
sys.exit(1 if bad else 0)
