#!/bin/bash
# round 5: is k_fused_pc bound by the marching wave's instruction ISSUE (one instruction of any kind per ~5 cycles and
# wave, tools/probes/valu_issue.hip)?  Ablation builds: 40 extra scalar / vector no-op instructions per row step, the
# march alone, the producer alone; then the dynamic instruction counts of the three from the SQ counters.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
V=$R/libjxl_amd/csrc/variants
q() { bash tools/quick.sh "$1" --no-pcie --frames-in-flight 1 --steps 200 --warmup 20 "${@:2}"; }
{
for rep in 1 2; do
  q ""
  for v in padS40 padV40 nofill nomarch; do q "JXLHIP_SO=$V/libjxl_hip_$v.so"; done
done
} 2>&1 | tee $O/issue_bound_bench.txt
{
for v in "" nofill nomarch; do
  so=""; [ -n "$v" ] && so="JXLHIP_SO=$V/libjxl_hip_$v.so"
  echo "=== variant [$v]"
  env $so PMC_TIMEOUT=200 bash tools/pmc_pass.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES" --no-pcie
  env $so PMC_TIMEOUT=200 bash tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH" --no-pcie
done
} 2>&1 | tee $O/issue_bound_counters.txt
