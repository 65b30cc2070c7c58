#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
rm -f $O/r02e.txt
for w in 7680 7672 7696 7744 7808 8192; do
  echo "== width $w" >> $O/r02e.txt
  python bench.py --no-cpu-baseline --steps 30 --warmup 5 --width $w 2>&1 | grep -o '"value": [0-9.]*\|kernel_ms": {[^}]*}' | tr '\n' ' ' >> $O/r02e.txt; echo >> $O/r02e.txt
done
for rh in 48 68 90 135 180 270 540; do
  echo "== RH $rh" >> $O/r02e.txt
  JXLHIP_FILTER_RH=$rh python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>&1 | grep -o 'filters": [0-9.]*' >> $O/r02e.txt
done
cat $O/r02e.txt
