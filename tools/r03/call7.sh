#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi.py tests/test_gpu_vs_reference.py -m gpu -q --tb=short -k "stripe or multi or fused or config3 or striped" 2>&1 | tail -30
B="python bench.py --no-cpu-baseline --no-pcie --steps 30 --warmup 5"
for fuse in 0 -1; do
JXLHIP_FUSE=$fuse timeout 300 python - <<'PY'
# one process, N stripes of the 16K frame on ONE device through jxlhip_decode_blocks / halo / decode_filters (libjxl_amd.stripes with world = 1 is
# the whole frame; so drive 4 stripe contexts by hand like tests do)
import os, sys, time, torch
sys.path.insert(0, ".")
from libjxl_amd import VarDctDecoder, synth
xs, ys = 15360, 8640
params, t = synth.synth_frame(xs, ys, mix=synth.MIX_D1, gab=True, epf_iters=1, device="cuda")
ysg = (ys + 255) // 256
n = 4
base, rem = divmod(ysg, n)
stripes, g0 = [], 0
for i in range(n):
    r = base + (1 if i < rem else 0); stripes.append((g0, r)); g0 += r
decs, outs = [], []
dq = None
for (g0, gr) in stripes:
    d = VarDctDecoder(0); d.begin_frame(dict(params, stripe_group_y0=g0, stripe_group_rows=gr))
    if dq is None: dq = d.default_dequant_tables()
    d.set_inputs(t, dq); decs.append(d); outs.append(d.alloc_output())
def step():
    for d in decs: d.decode_blocks()
    for i in range(n - 1):
        decs[i + 1].halo_import(0, decs[i].halo_export(1)); decs[i].halo_import(1, decs[i + 1].halo_export(0))
    for d, o in zip(decs, outs): d.decode_filters(o)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("16K, 4 stripes on one device, JXLHIP_FUSE=%s: %.3f ms/frame = %.1f Gpx/s" % (os.environ.get("JXLHIP_FUSE"), dt * 1e3, xs * ys / dt / 1e9))
PY
done
