#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02end_tests.txt 2>&1; tail -3 $O/r02end_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 | cut -c1-420
