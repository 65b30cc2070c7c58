#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
rm -f $O/r02d.txt
for so in "" a1b1 a1b2 a2b2 a1b4; do
 for res in 512 768; do
  for dbg in 0 4; do
   echo "== so=$so resident=$res dbg=$dbg" >> $O/r02d.txt
   JXLHIP_SO=${so:+$R/libjxl_amd/csrc/variants/libjxl_hip_$so.so} JXLHIP_FILTER_RESIDENT=$res JXLHIP_DEBUG=$dbg python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>&1 | grep -o 'filters": [0-9.]*' >> $O/r02d.txt
  done
 done
done
paste - - < $O/r02d.txt
# parity on the burst variant
JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_a1b4.so timeout 600 python -m pytest tests/test_gpu_vs_reference.py -m gpu -x -q 2>&1 | tail -n 3
