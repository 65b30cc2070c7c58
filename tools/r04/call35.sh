#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 | tee $O/r04_call35_smoke.txt
python bench.py > $O/r04_call35_bench.json 2> $O/r04_call35_bench.err; tail -c 300 $O/r04_call35_bench.err
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $O/r04_call35_tests.txt
