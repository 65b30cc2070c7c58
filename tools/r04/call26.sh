#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/r04/wp_variants.py current bmi2_clone no_clones 2>&1 | tee $O/r04_wp_variants_box2.txt
echo "== pipelined + staged"; JXLHIP_CODESTREAM_VERBOSE=1 python tools/r04/e2e_timeline.py 64 12 2>&1 | grep -v "DC-phase units" | tee $O/r04_e2e_staged.txt
timeout 900 python -m pytest tests -q -m gpu -k "codestream or djxl or extra or conformance or alpha or fuzz or soak" 2>&1 | tail -5 | tee $O/r04_call26_tests.txt
