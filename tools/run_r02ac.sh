#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -m gpu -x -q -k "fused or config or packed_8k" > $O/r02ac_tests.txt 2>&1; tail -3 $O/r02ac_tests.txt
for i in 1 2; do bash tools/kstats.sh "A=1" --config c3 --no-pcie --steps 30 --warmup 5 2>&1 | grep "value\|k_fused\|k_transform_r" ; done
bash tools/kstats.sh "A=1" --config c4 --no-pcie --steps 10 --warmup 3 2>&1 | grep "value\|k_fused\|k_transform_r"
bash tools/kstats.sh "A=1" --config c2 --no-pcie --steps 30 --warmup 5 2>&1 | grep "value\|k_fused\|k_transform_r"
