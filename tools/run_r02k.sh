#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/r02k_tests.txt 2>&1; tail -5 $O/r02k_tests.txt
for m in 0 1; do
  bash tools/kstats.sh "JXLHIP_MFMA=$m" --config c5 --no-pcie --steps 20 --warmup 5 > $O/r02k_c5_mfma$m.txt 2>&1
  cat $O/r02k_c5_mfma$m.txt
done
rm -f $O/r02k_pmc.txt
for m in 0 1; do
export JXLHIP_MFMA=$m
for set in "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES"; do
  echo "== JXLHIP_MFMA=$m $set" >> $O/r02k_pmc.txt
  bash $R/tools/pmc_pass.sh "$set" --config c5 --no-pcie >> $O/r02k_pmc.txt 2>&1
done
done
cut -c1-230 $O/r02k_pmc.txt | grep -v "^E2\|^W2"
