#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02aa.txt
for rep in 1 2; do
for v in w3 w2; do
  for res in 512 768; do
  bash tools/kstats.sh "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$v.so JXLHIP_FUSED_RESIDENT=$res" --config c3 --no-pcie --steps 30 --warmup 5 2>&1 | grep "env=\|value\|k_fused" | cut -c1-150 >> $O/r02aa.txt
  done
done
done
cat $O/r02aa.txt
