#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/r04/build_timing_lib.sh /tmp/libjxl_timing.so
JXLHIP_LIB=/tmp/libjxl_timing.so python tools/r04/dc_bench.py 3 2>&1 | grep -v "^\[" | tail -60 | tee $O/r04_dc_bench_box.txt
python tools/r04/dc_bench.py 5 2>&1 | tail -6 | tee -a $O/r04_dc_bench_box.txt
