#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
for i in 1 2 3; do python tools/r04/e2e_timeline.py 64 5 2>&1 | grep "rep  [01]:"; done | tee $O/r04_e2e_cold.txt
timeout 900 python -m pytest tests -q -m gpu -k "codestream or djxl or entropy or front_end or extra or multi" 2>&1 | tail -3
