#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vs_reference.py tests/test_gpu_parity.py -m gpu -x -q -k "config4 or dct32 or mfma" > $O/r02ae_tests.txt 2>&1; tail -6 $O/r02ae_tests.txt
for m in "A=1" "JXLHIP_MFMA=0"; do bash tools/kstats.sh "$m" --config c5 --no-pcie --steps 30 --warmup 5 2>&1 | grep "env=\|value\|kernel_ms\|k_"; done
