#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/r04/wp_variants.py current no_slp no_slp_no_cmovconv no_slp_O2 no_slp_no_ans bmi2_clone 2>&1 | tee $O/r04_wp_variants_box3.txt
