#!/bin/bash
# GPU box (round 6): where do k_transform_r's extra read bytes come from?  HBM bytes per launch of the
# phase-1 kernels (FETCH_SIZE / WRITE_SIZE passes, calibrated on a 1 GiB copy: tools/pmc_traffic.sh) on 8K frames of ONE
# strategy each, two-phase (JXLHIP_FUSE=0: every block through phase 1), against the bytes the class needs:
# coefficients 6 B/px + DC 12 B per covered block + 16 B per work item; writes 12 B/px.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
export JXLHIP_FUSE=0
for mix in ${MIXES:-0:1 4:1 6:1 7:1 5:1 10:1 18:1 19:1 2:1 13:1}; do
  tag=r06_over_$(echo $mix | tr : _)
  bash tools/pmc_traffic.sh $tag --no-pcie --no-e2e --steps 6 --warmup 2 --mix $mix > /dev/null 2>&1
  python - $mix $O/pmc_traffic_$tag.json <<'PY'
import json, sys
mix, path = sys.argv[1], sys.argv[2]
s = int(mix.split(":")[0])
CX = [1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8]
CY = [1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4]
try:
    d = json.load(open(path))["per_launch_detail"]
except Exception as ex:
    print(mix, "no data", ex); sys.exit(0)
px = 7680 * 4320
cells = px / 64
need_r = px * 6 + cells * 12 + cells / (CX[s] * CY[s]) * 16
need_w = px * 12
for k, v in d.items():
    if k.startswith("blocks") and v['read_bytes'] > 20e6:
        print(f"strategy {s:2d} ({CX[s]*8}x{CY[s]*8}) {k:14s} read {v['read_bytes']/1e6:7.1f} MB = {v['read_bytes']/need_r:5.2f}x of {need_r/1e6:6.1f}   write {v['write_bytes']/1e6:7.1f} MB = {v['write_bytes']/need_w:5.2f}x")
PY
done
