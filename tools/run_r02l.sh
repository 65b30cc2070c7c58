#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/r02l_tests.txt 2>&1; tail -15 $O/r02l_tests.txt
for g in 1 0; do
  bash tools/kstats.sh "A=1" --config c3 --epf 3 --gab $g --no-pcie --steps 20 --warmup 5 > $O/r02l_c3_epf3_gab$g.txt 2>&1
  cat $O/r02l_c3_epf3_gab$g.txt
done
bash tools/kstats.sh "JXLHIP_FILTERS=generic" --config c3 --epf 3 --no-pcie --steps 5 --warmup 2 > $O/r02l_c3_epf3_generic.txt 2>&1
cat $O/r02l_c3_epf3_generic.txt
