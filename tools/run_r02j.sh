#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mfma" > $O/r02j_tests.txt 2>&1; tail -5 $O/r02j_tests.txt
for m in 0 1; do
  bash tools/kstats.sh "JXLHIP_MFMA=$m" --config c5 --no-pcie --steps 20 --warmup 5 > $O/r02j_c5_mfma$m.txt 2>&1
  cat $O/r02j_c5_mfma$m.txt
done
