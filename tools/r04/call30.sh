#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py > $O/r04_call30_bench.json 2> $O/r04_call30_bench.err; tail -c 6000 $O/r04_call30_bench.json
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $O/r04_call30_tests.txt
