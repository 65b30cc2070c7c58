#!/bin/bash
# GPU box: diagnostic counters for the round-1 kernels (what bounds k_filters_fast / k_transform_*)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/r02a_counters_list.txt 2>&1
python $R/tools/membw.py > $O/r02a_membw.txt 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  echo "== $set" >> $O/r02a_pmc.txt
  bash $R/tools/pmc_pass.sh "$set" >> $O/r02a_pmc.txt 2>&1
  tail -3 /tmp/pmcx.log >> $O/r02a_pmc.txt
done
for dbg in 0 4 8 12; do
  echo "== JXLHIP_DEBUG=$dbg" >> $O/r02a_ablate.txt
  JXLHIP_DEBUG=$dbg python $R/bench.py --no-cpu-baseline --steps 50 --warmup 5 >> $O/r02a_ablate.txt 2>&1
done
