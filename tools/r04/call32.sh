#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/r04/dc_bench.py 3 2>&1 | tail -5 | tee $O/r04_dc_bench_box2.txt
python tools/r04/e2e_timeline.py 64 16 2>&1 | tee $O/r04_e2e_final.txt
python tools/r04/e2e_timeline.py 32 12 2>&1 | tee -a $O/r04_e2e_final.txt
timeout 900 python -m pytest tests -q -m gpu -k "codestream or djxl or entropy or front_end" 2>&1 | tail -3
