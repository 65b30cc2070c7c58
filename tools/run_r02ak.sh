#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02ak.txt
for sz in "3840 2160" "1920 1080" "7680 1080"; do
  w=${sz% *}; h=${sz#* }
  for bs in 1 2; do
    JXLHIP_FUSE=0 JXLHIP_BLOCK_STREAMS=$bs python bench.py --config c3 --width $w --height $h --no-pcie --no-cpu-baseline --steps 50 --warmup 5 > /tmp/b.log 2>&1
    echo "${w}x${h} streams=$bs $(grep -o '"value": [0-9.]*' /tmp/b.log) $(grep -o 'kernel_ms.: {[^}]*}' /tmp/b.log)" >> $O/r02ak.txt
  done
done
cat $O/r02ak.txt
