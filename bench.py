#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the VarDCT decode back-end on MI355X.

One "step" = one pass of the hot path (dequant -> inverse transforms ->
Gaborish -> EPF1 -> XYB->linear RGB) over one synthetic frame whose quantized
coefficients and side info are already resident in HBM; output stays in HBM.

  N = 1 : BASELINE.json configs[2]: 7680x4320 RGB, d1.0-like (Gaborish + EPF1),
          int16 coefficients, d1.0/e7-like strategy mix.
  N > 1 : weak scaling: a 7680 x (4320*N) frame split into N group-row stripes
          (one rank per GPU), halo rows exchanged with the two neighbours over
          RCCL between the two phases; each rank's output stripe stays in its HBM
          (pass --gather to also time the collection on rank 0).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(xsize, ysize, coeff_bytes):
    """B = Wp*Hp*3*sizeof(coef) + Nblk*18 + W*H*12."""
    wp, hp = (xsize + 7) // 8 * 8, (ysize + 7) // 8 * 8
    return wp * hp * 3 * coeff_bytes + (wp * hp // 64) * 18 + xsize * ysize * 12


def cpu_baseline(args):
    """libjxl's own CPU path timed on the host cores on a bounded sample of the
    same workload: oracle/_ref = the reference decoder sources compiled in place
    (DecodeGroupForRoundtrip + LowMemoryRenderPipeline, the executor djxl uses),
    threaded over groups.  NOTE the build: Highway is not vendored in the
    reference tree, so the SIMD layer is oracle/hwy_shim (ONE lane per vector,
    scalar code) -- this is libjxl's algorithm and code, not its AVX2/AVX-512
    speed.  Falls back to the C restatement (kind "port") without oracle/_ref."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frames
    import oracle
    from libjxl_amd import synth
    w, h = args.cpu_sample
    cores = os.cpu_count() or 1
    _, _, fr = frames.make_case(w, h, mix=synth.MIX_D1, gab=True, epf_iters=1)
    use_ref = oracle.ref_available()
    run = (lambda: fr.decode_ref(threads=cores)) if use_ref else (lambda: fr.decode(threads=cores))
    run()  # warm
    reps, t = 0, 0.0
    while reps < 2 or (t < 10.0 and reps < 12):
        t0 = time.perf_counter()
        run()
        t += time.perf_counter() - t0
        reps += 1
    what = ("libjxl reference sources (lib/jxl, DecodeGroupForRoundtrip + LowMemoryRenderPipeline) built with "
            "the single-lane Highway shim (scalar; not the AVX2/AVX-512 build)") if use_ref else \
        "oracle/ C restatement (libjxl reference library not available)"
    return {"value": round(w * h * reps / t / 1e6, 2), "unit": "Mpixels/s", "cores": cores,
            "kind": "reference" if use_ref else "port",
            "sample": f"{w}x{h} d1.0-like frame (Gaborish+EPF1), {reps} reps, {cores} threads over groups; {what}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--gab", type=int, default=1)
    ap.add_argument("--epf", type=int, default=1)
    ap.add_argument("--mix", default="d1",
                    help="d1 | dct8 | dct32 | all, or an explicit area mix 'strategy:share,...' (e.g. 18:1 = all 64x64)")
    ap.add_argument("--coeff32", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also gather stripes on rank 0 each step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calib-copy", action="store_true",
                    help="run one known-size (1 GiB) device copy so PMC passes can be calibrated")
    ap.add_argument("--cpu-sample", type=int, nargs=2, default=[4096, 2160])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from libjxl_amd import VarDctDecoder, stripes, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the VarDCT back-end has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    mix = {"d1": synth.MIX_D1, "dct8": synth.MIX_DCT8, "dct32": synth.MIX_DCT32,
           "all": synth.MIX_ALL}.get(args.mix) or {int(k): float(v) for k, v in
                                                    (kv.split(":") for kv in args.mix.split(","))}
    xs, ys = args.width, args.height * world
    params, t = synth.synth_frame(xs, ys, mix=mix, gab=bool(args.gab), epf_iters=args.epf,
                                  device=f"cuda:{local}", coeff_type=int(args.coeff32))
    dec = VarDctDecoder(local)
    sd = stripes.StripeDecoder(dec, params, rank, world)
    dq = dec.default_dequant_tables()
    dec.set_inputs(t, dq)
    out = dec.alloc_output()
    rows = [b - a for a, b in sd.rows]

    def step():
        sd.decode(out)
        if args.gather and world > 1:
            stripes.gather_stripes(out, rows, rank, world)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.calib_copy:
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        del a, b
    for _ in range(args.warmup):
        step()
    dec.sync()  # also surfaces stream errors
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel device time with HIP events on the launch stream (own pass)
    dec.profile(True)
    for _ in range(args.steps):
        dec.decode_blocks()
        dec.decode_filters(out)
    prof = dec.profile_read()
    dec.profile(False)
    kern = {k: round(ms / max(n, 1), 4) for k, (ms, n) in prof.items()}

    if rank == 0:
        px = xs * ys
        ms_step = dt / args.steps * 1e3
        value = px / (dt / args.steps) / 1e6
        # dominant SINGLE kernel of this rank's stripe: the "blocks" slot is a span
        # over the transform launches (k_prepare, k_transform_8, k_transform_r; the
        # longest of them is ~0.4x the filter kernel in profiles/*_kernel_stats.csv),
        # so the roofline is quoted on the fused Gaborish+EPF+XYB kernel, one launch
        # per frame here.
        dom = "filters" if "filters" in kern else max(kern, key=kern.get)
        y0, y1 = sd.rows[rank]
        b_alg = algorithmic_bytes(xs, y1 - y0, 4 if args.coeff32 else 2)
        achieved = b_alg / (kern[dom] * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf) and world == 1 and (xs, ys) == (7680, 4320):
            try:
                traffic = json.load(open(tf)).get(dom)
            except Exception:
                traffic = None
        line = {
            "metric": "Mpixels/s decode (VarDCT d1.0, 8K RGB)", "value": round(value, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{xs}x{ys} RGB VarDCT d1.0-like frame, gab={args.gab} "
                                   f"epf_iters={args.epf}, {'int32' if args.coeff32 else 'int16'} "
                                   f"coefficients, strategy mix {args.mix}, linear RGB f32 out",
                       "stripes": world, "halo_rows": dec.halo_rows(),
                       "gather_in_step": bool(args.gather),
                       "kernel_ms": kern},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel": "k_filters_fast" if dom == "filters" else dom,
                         "algorithmic_bytes_per_launch": b_alg},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
