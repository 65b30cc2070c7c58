#!/usr/bin/env python3
"""Round 6 experiment driver: one frame at a time on a fresh context per environment setting (the switches are read
when a context is created); every setting decodes the same frame, is compared with the first one's pixels, timed like
bench.py's `value`, and followed by a per-kernel HIP-event pass.
usage: env_sweep.py [--mix d1|real4k|..] [--gab 1 --epf 1] --envs 'A=1,B=2;C=3;...'   (an empty entry = the defaults)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mix", default="d1")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--gab", type=int, default=1)
    ap.add_argument("--epf", type=int, default=1)
    ap.add_argument("--used-acs", action="store_true", help="hand the mix's used_acs mask to the decoder")
    ap.add_argument("--no-used-acs", action="store_true", help="used_acs = 0 (unknown): everything rides in the merged phase-1 launch")
    ap.add_argument("--envs", default="")
    args = ap.parse_args()
    import torch
    import bench
    from libjxl_amd import VarDctDecoder, synth
    xs, ys = args.width, args.height
    mix = bench.resolve_mix(args.mix)
    params, t = synth.synth_frame(xs, ys, mix=mix, gab=bool(args.gab), epf_iters=args.epf, device="cuda:0")
    if args.used_acs:
        m = 0
        for s in mix:
            m |= 1 << int(s)
        params["used_acs"] = m
    if args.no_used_acs:
        params["used_acs"] = 0
    ref = None
    touched = set()
    for spec in args.envs.split(";"):
        for k in touched:
            os.environ.pop(k, None)
        touched = set()
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
            touched.add(k)
        dec = VarDctDecoder(0)
        dec.begin_frame(params)
        dq = dec.default_dequant_tables()
        dec.set_inputs(t, dq)
        out = dec.alloc_output()
        for _ in range(3):
            dec.decode_frame(out)
        dec.sync()
        if ref is None:
            ref = out.clone()
        err = float((out - ref).abs().max())
        ms = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.06:
                for _ in range(8):
                    dec.decode_frame(out)
                dec.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dec.decode_frame(out)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) / args.steps * 1e3)
        ms.sort()
        med = ms[len(ms) // 2]
        dec.profile(True)
        for _ in range(10):
            dec.decode_frame(out)
        prof = dec.profile_read()
        dec.profile(False)
        kern = {k: round(v / max(n, 1) * 1e3, 1) for k, (v, n) in prof.items()}
        print(json.dumps(dict(env=spec, ms=round(med, 4), best=round(ms[0], 4), gpx=round(xs * ys / med / 1e6, 1),
                              max_abs_diff_vs_first=err, kernel_us=kern)), flush=True)
        dec.close()
        del dec, out


if __name__ == "__main__":
    main()
