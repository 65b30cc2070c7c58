#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python tools/r04/mall_probe.py 2>&1 | tee $O/r04_mall_probe.txt
for fif in 1 3; do for br in 0 1 2 4; do
  echo "two-phase, frames in flight $fif, band rows $br" 
  JXLHIP_FUSE=0 JXLHIP_BAND_ROWS=$br python bench.py --steps 60 --warmup 10 --no-e2e --no-cpu-baseline --no-pcie --frames-in-flight $fif 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/r04_band_rows.txt
