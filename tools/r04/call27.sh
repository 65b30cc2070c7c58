#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
JXLHIP_CODESTREAM_VERBOSE=1 python tools/r04/e2e_timeline.py 64 30 2>&1 | grep -v "DC-phase units\|from the headers" | tee $O/r04_e2e_waits.txt
