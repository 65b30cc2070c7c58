#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
rm -f $O/r02c.txt
for so in "" ahead3 ahead4; do
 for res in 512 768 1024; do
  for dbg in 0 4; do
   echo "== so=$so resident=$res dbg=$dbg" >> $O/r02c.txt
   JXLHIP_SO=${so:+$R/libjxl_amd/csrc/variants/libjxl_hip_$so.so} JXLHIP_FILTER_RESIDENT=$res JXLHIP_DEBUG=$dbg python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>&1 | grep -o 'kernel_ms": {[^}]*}' >> $O/r02c.txt
  done
 done
done
cat $O/r02c.txt
