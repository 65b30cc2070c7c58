#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pipelined"; JXLHIP_CODESTREAM_VERBOSE=1 python tools/r04/e2e_timeline.py 64 10 2>&1 | grep -v "DC-phase units" | tee $O/r04_e2e_pipelined.txt
echo "== not pipelined"; JXLHIP_NO_PIPELINE=1 python tools/r04/e2e_timeline.py 64 10 2>&1 | tee $O/r04_e2e_not_pipelined.txt
echo "== pipelined, 128 threads"; python tools/r04/e2e_timeline.py 128 10 2>&1 | tee $O/r04_e2e_pipelined128.txt
timeout 900 python -m pytest tests -q -m gpu -k "codestream or djxl or extra or conformance or alpha or fuzz or soak" 2>&1 | tail -5 | tee $O/r04_call23_tests.txt
