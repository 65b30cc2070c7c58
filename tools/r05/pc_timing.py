"""GPU box: where do the two waves of k_fused_pc spend their time?  Runs the -DJXLHIP_PC_TIMING experiment build
(libjxl_amd/csrc/variants/libjxl_hip_pctiming.so) on the c3 frame: every wave sums the shader-clock ticks it spends waiting
for its own memory / LDS operations in front of a barrier and inside s_barrier waiting for the other wave; the sums land in
the first words of inv_sigma (pixels near the frame's corner are garbage in this build).
usage: JXLHIP_SO=.../libjxl_hip_pctiming.so python tools/r05/pc_timing.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from libjxl_amd import VarDctDecoder, synth
import bench
mix = bench.resolve_mix("d1")
for name, m in (("c3 (d1 mix)", mix), ("genuine-content mix", bench.resolve_mix("real4k"))):
    params, t = synth.synth_frame(7680, 4320, mix=m, gab=True, epf_iters=1, device="cuda:0")
    dec = VarDctDecoder(0)
    dec.begin_frame(params)
    dec.set_inputs(t, dec.default_dequant_tables())
    out = dec.alloc_output()
    for _ in range(30):
        dec.decode_frame(out)
    dec.sync()
    raw = dec.sigma().cpu().numpy().reshape(-1)[:16].view(np.uint64)
    for role, label in ((0, "marching wave"), (1, "producing wave")):
        total, wmem, wbar, n = [int(x) for x in raw[4 * role:4 * role + 4]]
        if n:
            print(f"{name}: {label}: {n} waves, run {total / n:9.0f} ticks; waiting for own memory / LDS in front of a barrier "
                  f"{100.0 * wmem / total:5.1f} %, inside s_barrier {100.0 * wbar / total:5.1f} %, computing {100.0 * (total - wmem - wbar) / total:5.1f} %")
    # placement: which waves shared a SIMD?  HW_ID (gfx9): wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]
    words = dec.sigma().cpu().numpy().reshape(-1)[64:64 + 8 * 1536 * 2 + 64].view(np.uint32)
    import collections
    simds = collections.defaultdict(list)
    for wg in range(1536):
        for role in (0, 1):
            hw, xcc, run, wbar = [int(words[4 * (2 * wg + role) + k]) for k in range(4)]
            if hw == 0 and xcc == 0:
                continue
            key = (xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3)
            simds[key].append((role, run, wbar))
    comp = collections.Counter((sum(1 for r in v if r[0] == 0), sum(1 for r in v if r[0] == 1)) for v in simds.values())
    print(f"{name}: SIMDs by (marching waves, producing waves) resident on them:", dict(sorted(comp.items())), "on", len(simds), "SIMDs")
    for kind in ((2, 1), (1, 2)):
        runs = [r[1] - r[2] for v in simds.values() if (sum(1 for r in v if r[0] == 0), sum(1 for r in v if r[0] == 1)) == kind
                for r in v if r[0] == 0]
        if runs:
            print(f"{name}: marching waves on SIMDs with {kind[0]} marches + {kind[1]} producers: {len(runs)} waves, run minus barrier "
                  f"time mean {np.mean(runs):9.0f} ticks (min {np.min(runs)}, max {np.max(runs)})")
    dec.close()
