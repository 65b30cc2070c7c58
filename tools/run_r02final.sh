#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/r02final_smoke.txt 2>&1; tail -5 $O/r02final_smoke.txt
( time python bench.py ) > $O/r02final_bench.txt 2>&1; tail -4 $O/r02final_bench.txt | cut -c1-1800
