#!/bin/bash
# round 4, GPU call 6: the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 | tee $O/r04_gpu_tests.txt
