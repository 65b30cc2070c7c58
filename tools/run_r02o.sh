#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/r02o_tests.txt 2>&1; tail -25 $O/r02o_tests.txt
