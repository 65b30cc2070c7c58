#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02z_tests.txt 2>&1; tail -12 $O/r02z_tests.txt
python bench.py --no-pcie --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | tail -1 | cut -c1-300
