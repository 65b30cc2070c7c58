#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02y.txt
for rep in 1 2; do
for v in "" "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_r2.so"; do
  for a in "--config c3" "--config c3 --mix real4k" "--config c2 --gab 1 --epf 1"; do
    bash tools/kstats.sh "A=1 $v" $a --no-pcie --steps 30 --warmup 5 2>&1 | grep "env=\|value\|k_transform_r" >> $O/r02y.txt
  done
done
done
cat $O/r02y.txt
