#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/r04/e2e_timeline.py 64 14 2>&1 | tee $O/r04_e2e_reps.txt
JXLHIP_CODESTREAM_VERBOSE=1 python tools/r04/e2e_timeline.py 64 8 2>&1 | tee $O/r04_e2e_timeline.txt
python tools/r04/e2e_timeline.py 32 8 2>&1 | tee $O/r04_e2e_reps32.txt
python tools/r04/e2e_timeline.py 128 8 2>&1 | tee $O/r04_e2e_reps128.txt
