#!/usr/bin/env python3
"""djxl_hip.py IN.jxl OUT.{pfm,npy,ppm,pam} [--threads N] [--reps K]

Decodes a .jxl file (container or bare codestream, one VarDCT still frame) on an MI355X through
jxlhip_decode_codestream (include/jxl_hip_codestream.h) and writes the pixels the way djxl does for these
extensions (tools/djxl_main.cc, lib/extras/enc/pnm.cc):
  .pfm  linear-light float RGB, bottom-up rows, little endian (scale -1.0)
  .npy  float32 [H, W, 3] linear RGB (what tools/conformance/conformance.py reads, conformance.py:34-66)
  .ppm  8-bit RGB               .pam  8-bit RGBA (the image's alpha channel, opaque without one)
        -- both in the image's original colour encoding (its transfer function over its primaries; an ICC
           original: linear sRGB, like djxl without a CMS)
Streams outside the back-end (Modular frames, patches / noise, animation ...) exit with status 3 and the error text so that a
wrapper can fall back to libjxl's djxl.  Prints Mpx/s of the decode call like djxl's SpeedStats."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--reps", type=int, default=1)
    a = ap.parse_args()
    import torch
    from libjxl_amd import VarDctDecoder, abi
    L = abi.load_library()
    blob = open(a.input, "rb").read()
    info = abi.CodestreamInfo()
    rc = L.jxlhip_codestream_basic_info(blob, len(blob), C.byref(info))
    if rc:
        sys.stderr.write(f"djxl_hip: {a.input}: {L.jxlhip_status_string(rc).decode()}\n")
        sys.exit(3 if rc == -7 else 1)
    ext = os.path.splitext(a.output)[1].lower()
    packed = ext in (".ppm", ".pam")
    R = C.CDLL(abi.runner_library_path())
    R.JxlThreadParallelRunnerCreate.restype = C.c_void_p
    R.JxlThreadParallelRunnerCreate.argtypes = [C.c_void_p, C.c_size_t]
    R.JxlThreadParallelRunnerDestroy.argtypes = [C.c_void_p]
    pool = R.JxlThreadParallelRunnerCreate(None, a.threads) if a.threads else None
    runner = C.cast(R.JxlThreadParallelRunner, C.c_void_p) if a.threads else None
    dec = VarDctDecoder(0)
    w, h = info.xsize, info.ysize
    if info.orientation >= 5:  # display orientation like djxl (JXLHIP_OUT_UNDO_ORIENTATION): transposed frame
        w, h = h, w
    UNDO = 0x100
    if packed:
        nc = 4 if ext == ".pam" else 3
        # 8-bit samples in the ORIGINAL colour encoding, like djxl: the original's transfer function over pixels in the
        # original's primaries (jxlhip_decode_codestream adapts the opsin inverse; the info struct names the rest)
        if info.gamma > 0:
            tf, par = 4, info.gamma
        else:
            tf, par = {8: (0, 0.0), 13: (1, 0.0), 16: (2, info.intensity_target), 1: (3, 0.0), 17: (4, 1 / 2.6),
                       18: (5, info.intensity_target)}.get(info.transfer_function, (1, 0.0))
        fmt = abi.OutputFormat(tf, 1, nc, 8, 0, par, info.luminances)
        out = torch.empty((h, w, nc), dtype=torch.uint8, device="cuda")
        args = (2 | UNDO, C.byref(fmt), out.data_ptr(), w * nc, 0)
    else:
        out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        args = (1 | UNDO, None, out.data_ptr(), w * 12, 0)
    best = None
    for _ in range(max(1, a.reps)):
        t0 = time.perf_counter()
        rc = L.jxlhip_decode_codestream(dec.ctx, runner, pool, blob, len(blob), *args, C.byref(info))
        dt = time.perf_counter() - t0
        if rc:
            sys.stderr.write(f"djxl_hip: {a.input}: {L.jxlhip_status_string(rc).decode()}: "
                             f"{L.jxlhip_last_error(dec.ctx).decode()}\n")
            sys.exit(3 if rc == -7 else 1)
        best = dt if best is None else min(best, dt)
    px = out.cpu().numpy()
    if ext == ".npy":
        np.save(a.output, px)
    elif ext == ".pfm":
        with open(a.output, "wb") as f:
            f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
            f.write(np.ascontiguousarray(px[::-1]).astype("<f4").tobytes())
    elif ext == ".ppm":
        with open(a.output, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (w, h))
            f.write(px.tobytes())
    elif ext == ".pam":
        with open(a.output, "wb") as f:
            f.write(b"P7\nWIDTH %d\nHEIGHT %d\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n" % (w, h))
            f.write(px.tobytes())
    else:
        sys.exit("output must be .pfm, .npy, .ppm or .pam")
    print(f"{w} x {h}, {w * h / best / 1e6:.1f} MP/s [{a.reps} reps, {a.threads} threads], "
          f"{info.num_groups} groups, {info.num_passes} pass(es), coefficients "
          f"{'int16' if info.coeff_type == 0 else 'int32'}, epf_iters {info.epf_iters} gab {info.gab}")
    dec.close()
    if pool:
        R.JxlThreadParallelRunnerDestroy(pool)


if __name__ == "__main__":
    main()
