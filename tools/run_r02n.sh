#!/bin/bash
# fused vs two-phase for every stage list the fused kernel takes (8K d1.0 mix), one box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02n.txt
for ge in "0 0" "1 0" "0 1" "1 1" "0 2" "1 2"; do
  set -- $ge
  for fu in 1 0; do
    echo "== gab=$1 epf=$2 JXLHIP_FUSE=$fu" >> $O/r02n.txt
    JXLHIP_FUSE=$fu python bench.py --config c3 --gab $1 --epf $2 --no-pcie --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|kernel_ms.: {[^}]*}' | tr '\n' ' ' >> $O/r02n.txt
    echo >> $O/r02n.txt
  done
done
cat $O/r02n.txt
