#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1700 python tools/soak.py 140 97 2>&1 | tail -4 | tee $O/r04_soak3.log
