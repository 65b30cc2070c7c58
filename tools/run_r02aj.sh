#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02aj.txt
for sz in "5120 2880" "6144 3456" "7680 3200"; do
  w=${sz% *}; h=${sz#* }
  for fu in 1 0; do
    JXLHIP_FUSE=$fu python bench.py --config c3 --width $w --height $h --no-pcie --no-cpu-baseline --steps 30 --warmup 5 > /tmp/b.log 2>&1
    echo "${w}x${h} JXLHIP_FUSE=$fu $(grep -o '"value": [0-9.]*' /tmp/b.log) $(tail -1 /tmp/b.log | cut -c1-80)" >> $O/r02aj.txt
  done
done
cat $O/r02aj.txt
