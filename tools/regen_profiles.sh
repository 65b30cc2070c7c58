#!/bin/bash
# GPU box: regenerates the round's evidence from the CURRENT tree in one go -- for every BASELINE config (c1..c5), the
# real-content mix and epf_iters = 3: the bench line, the rocprofv3 --kernel-trace --stats summary of the same command
# and the per-launch HBM traffic (FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/<tag>_<cfg>_*; plus the full default
# bench line (cpu_baseline, pcie_inclusive) and the matrix-core counters of the c5 / c3 kernels.
# usage: tools/regen_profiles.sh <tag> [commit id]       then, in the build container: tools/collect_profiles.sh <tag>
tag=${1:-r03}; commit=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "$commit" > $O/${tag}_commit.txt
run() { cfg=$1; shift; bash tools/profile_round.sh ${tag}_$cfg "$@" > $O/${tag}_${cfg}_summary.txt 2>&1; tail -12 $O/${tag}_${cfg}_summary.txt | cut -c1-170; }
run c3
run c1 --config c1
run c2 --config c2
run c4 --config c4
run c5 --config c5
run real8k --mix real4k
run c3_epf3 --epf 3
timeout 1200 python bench.py > $O/${tag}_c3_bench_full.json 2> /dev/null
# (frames in flight: the default bench line above carries `frames_in_flight` with private inputs per context)
# matrix cores: the shipped c5 kernel (k_transform_mfma32<EMIT>) and the c3 step (no MFMA instruction in it)
for cfg in c5 c3; do
  extra=""; [ $cfg = c5 ] && extra="--config c5"
  bash tools/pmc_pass.sh "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" --no-pcie $extra > $O/${tag}_${cfg}_mfma_counters.txt 2>&1
  bash tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" --no-pcie $extra >> $O/${tag}_${cfg}_mfma_counters.txt 2>&1
  tail -8 $O/${tag}_${cfg}_mfma_counters.txt | cut -c1-200
done
