#!/bin/bash
# MFMA 32x32 IDCT (kernels_mfma.hip) vs the row-per-lane butterflies: parity, c5 kernel times, MFMA counters
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfma" > $O/r02i_tests.txt 2>&1; tail -5 $O/r02i_tests.txt
for m in 0 1; do
  bash tools/kstats.sh "JXLHIP_MFMA=$m" --config c5 --no-pcie --steps 20 --warmup 5 > $O/r02i_c5_mfma$m.txt 2>&1
  cat $O/r02i_c5_mfma$m.txt
done
for m in 0 1; do
  bash tools/kstats.sh "JXLHIP_MFMA=$m" --config c3 --no-pcie --steps 20 --warmup 5 > $O/r02i_c3_mfma$m.txt 2>&1
  cat $O/r02i_c3_mfma$m.txt
done
export JXLHIP_MFMA=1
for set in "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  echo "== $set" >> $O/r02i_pmc.txt
  bash $R/tools/pmc_pass.sh "$set" --config c5 --no-pcie >> $O/r02i_pmc.txt 2>&1
done
cut -c1-230 $O/r02i_pmc.txt | grep -v "^E2\|^W2"
