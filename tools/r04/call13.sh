#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -6 | tee $O/r04_gpu_tests.txt
bash tools/regen_profiles.sh r04 e28ae4535384
