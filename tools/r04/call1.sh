#!/bin/bash
# round 4, GPU call 1: the matrix-core tile producer (JXLHIP_FUSED_TILES) -- parity of the fused tests, then the c3 / real
# mix / c2 step with the producer on and off, and the load-slot variants
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "fused or tile" 2>&1 | tail -25 > $O/r04_call1_tests.txt
cat $O/r04_call1_tests.txt
q() { bash tools/quick.sh "$1" --no-pcie --steps 200 --warmup 20 "${@:2}"; }
{
for rep in 1 2; do
  q "JXLHIP_FUSED_TILES=1"; q "JXLHIP_FUSED_TILES=0"
done
q "JXLHIP_FUSED_TILES=1" --mix real4k; q "JXLHIP_FUSED_TILES=0" --mix real4k
q "JXLHIP_FUSED_TILES=1" --config c2; q "JXLHIP_FUSED_TILES=0" --config c2
q "JXLHIP_FUSED_TILES=1" --config c4; q "JXLHIP_FUSED_TILES=0" --config c4
for v in slots4 slots3 slots8; do
  [ -f $R/libjxl_amd/csrc/variants/libjxl_hip_$v.so ] && q "JXLHIP_FUSED_TILES=1 JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$v.so"
done
for rh in 104 136 200 280; do q "JXLHIP_FUSED_TILES=1 JXLHIP_FUSED_PC_RH=$rh"; done
} 2>&1 | tee $O/r04_call1_bench.txt
