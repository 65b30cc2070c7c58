#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02end2_tests.txt 2>&1; tail -3 $O/r02end2_tests.txt
for c in c1 c2; do bash tools/profile_round.sh r02_$c --config $c > $O/r02end2_$c.txt 2>&1; grep -o '"value": [0-9.]*' $O/r02_${c}_bench.json; done
python bench.py 2>/dev/null | tail -1 | cut -c1-200
