#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
for fu in 1 0; do echo "== JXLHIP_FUSE=$fu"; JXLHIP_FUSE=$fu python tools/packed_bench.py 2>&1 | tail -6; done > $O/r02ah.txt 2>&1
cat $O/r02ah.txt
