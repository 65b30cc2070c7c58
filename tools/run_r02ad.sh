#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02ad.txt
for rep in 1 2 3; do
for v in plain ldsonly issuelds; do
  bash tools/kstats.sh "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$v.so" --config c3 --no-pcie --steps 30 --warmup 5 2>&1 | grep "env=\|value\|k_fused" | cut -c1-120 >> $O/r02ad.txt
done
done
cat $O/r02ad.txt
