#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "orientation or codestream or seam or multi" > $O/r02z2_tests.txt 2>&1; tail -5 $O/r02z2_tests.txt
