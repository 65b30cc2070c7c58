#!/bin/bash
# round 5: rows per window chunk / store kind / producer priority with the leaner march + producer
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
V=$R/libjxl_amd/csrc/variants
q() { bash tools/quick.sh "$1" --no-pcie --frames-in-flight 1 --steps 200 --warmup 20 "${@:2}"; }
{
q ""; q ""
for v in plainstore prio0; do q "JXLHIP_SO=$V/libjxl_hip_$v.so"; done
for rh in 72 104 136 168 200 240 280 360; do q "JXLHIP_FUSED_PC_RH=$rh"; done
q ""
for v in plainstore prio0; do q "JXLHIP_SO=$V/libjxl_hip_$v.so"; done
} 2>&1 | tee $O/sweep_bench.txt
