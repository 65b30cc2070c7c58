#!/bin/bash
# round 4, GPU call 10: s_setprio experiments on k_fused_pc (march ahead of / behind the producing waves)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
one() { env $1 python bench.py --no-cpu-baseline --no-pcie --steps 300 --warmup 30 --frames-in-flight $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-60s in flight $2: %8.1f Gpx/s  %s' % ('$1'[-60:], d['value']/1e3, d['config']['kernel_ms']))"; }
{
for rep in 1 2; do
for v in "" marchprio prodprio; do
  e="X=1"; [ -n "$v" ] && e="JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$v.so"
  one "$e" 1; one "$e" 3
done; done
} 2>&1 | tee $O/r04_setprio.txt
