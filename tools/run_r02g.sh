#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused" > $O/r02g_pytest.txt 2>&1; tail -n 15 $O/r02g_pytest.txt
for fuse in 1 0; do
  echo "== fuse $fuse"
  JXLHIP_FUSE=$fuse timeout 300 python bench.py --no-cpu-baseline --no-pcie --steps 40 --warmup 5 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|kernel_ms": {[^}]*}' | tr '\n' ' '; echo
done
