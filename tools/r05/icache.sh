#!/bin/bash
# round 5: do the long straight-line units of k_transform_r (32- and 64-point classes, all roles in one launch) starve on
# instruction fetch?  SQ instruction-cache and wait counters on the genuine-stream shares and the d1 mix.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
{
for m in real4k d1; do
  echo "=== mix $m"
  PMC_TIMEOUT=200 bash tools/pmc_pass.sh "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --no-pcie --mix $m
  PMC_TIMEOUT=200 bash tools/pmc_pass.sh "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ" --no-pcie --mix $m
  PMC_TIMEOUT=200 bash tools/pmc_pass.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU" --no-pcie --mix $m
done
} 2>&1 | tee $O/icache.txt
