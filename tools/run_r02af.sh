#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02af_tests.txt 2>&1; tail -3 $O/r02af_tests.txt
bash tools/profile_round.sh r02_c5 --config c5 > $O/r02af_c5.txt 2>&1; tail -n 12 $O/r02af_c5.txt | cut -c1-170
bash tools/profile_round.sh r02_c3 --config c3 > $O/r02af_c3.txt 2>&1; tail -n 12 $O/r02af_c3.txt | cut -c1-170
