#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_codestream.py -q -m gpu -k "damaged_streams_end" 2>&1 | tail -15 | tee $O/r04_call49_tests.txt
