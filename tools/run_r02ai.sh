#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for i in 1 2; do
echo "two-phase:"; python tools/packed_one.py 2>&1 | tail -1
echo "fused fixed-format:"; JXLHIP_FUSE=1 JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_pk8.so python tools/packed_one.py 2>&1 | tail -1
done
