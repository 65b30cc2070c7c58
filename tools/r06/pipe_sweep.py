#!/usr/bin/env python3
"""RECORD OF AN EXPERIMENT THAT WAS NOT SHIPPED: the JXLHIP_PIPE_* switches this script sets exist only in a library built
with tools/r06/band_pipeline.patch, a diff against commit 353abba (the tree before round 6's pruning: it no longer applies to
HEAD).  Results: profiles/r06_band_pipeline_sweep.txt, profiles/r06_band_pipeline_timeline.txt, DESIGN.md section 4b.

Round 6 experiment driver: the banded frame (phase 1 of band b + 1 beside the fused kernel of band b) against the
one-band frame, over the knobs of context.hip's DecodeFramePipelined.  Every configuration is a fresh context (the
switches are read when a context is created), decodes the SAME frame, must be bit-equal to the one-band output, and is
timed like bench.py's `value`: one context, one frame at a time, after a settle phase.
usage: pipe_sweep.py [--mix d1|real4k] [--steps 40] [--configs 'bands,streams,rows,per_cu;...']"""
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
    ap.add_argument("--configs", default=None)
    ap.add_argument("--env", default="", help="extra KEY=VAL,... applied to every configuration")
    args = ap.parse_args()
    import torch
    import bench
    from libjxl_amd import VarDctDecoder, synth
    xs, ys = args.width, args.height
    params, t = synth.synth_frame(xs, ys, mix=bench.resolve_mix(args.mix), gab=True, epf_iters=1, device="cuda:0")
    for kv in filter(None, args.env.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    if args.configs:
        cfgs = [tuple(int(x) for x in c.split(",")) for c in args.configs.split(";")]
    else:
        cfgs = [(0, 1, 208, 0), (0, 1, 208, 5), (0, 1, 208, 4), (0, 1, 208, 3)]
        for nb in (2, 3, 4, 6):
            for ns in (1, 2, 3):
                for pc in (0, 4):
                    cfgs.append((nb, ns, 208, pc))
        cfgs += [(4, 2, 144, 0), (4, 2, 288, 0), (3, 2, 288, 4), (8, 2, 208, 0), (8, 3, 136, 0)]
    ref = None
    rows = []
    for nb, ns, pr, pc in cfgs:
        os.environ["JXLHIP_PIPE_BANDS"] = str(nb)
        os.environ["JXLHIP_PIPE_STREAMS"] = str(ns)
        os.environ["JXLHIP_PIPE_ROWS"] = str(pr)
        os.environ["JXLHIP_PIPE_PER_CU"] = str(pc)
        dec = VarDctDecoder(0)
        dec.begin_frame(params)
        dq = dec.default_dequant_tables()
        dec.set_inputs(t, dq)
        out = dec.alloc_output()
        dec.decode_frame(out)
        dec.sync()
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        out.zero_()
        for _ in range(3):  # frames back to back: the counter-block rotation, the cross-frame hazards
            dec.decode_frame(out)
        dec.sync()
        same = same and bool(torch.equal(out, ref))
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
        row = dict(bands=nb, streams=ns, rows=pr, per_cu=pc, ms=round(med, 4), best=round(ms[0], 4),
                   gpx=round(xs * ys / med / 1e6, 1), bit_equal=same)
        rows.append(row)
        print(json.dumps(row), flush=True)
        dec.close()
        del dec, out
    base = rows[0]["ms"]
    best = min(rows, key=lambda r: r["ms"])
    print(json.dumps(dict(summary=True, mix=args.mix, base_ms=base, best=best, all_bit_equal=all(r["bit_equal"] for r in rows))))


if __name__ == "__main__":
    main()
