#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
for wh in "3840 2160" "2560 1440" "1920 1080" "5120 2880"; do set -- $wh
  for fuse in 0 1; do for fif in 1 3; do
    JXLHIP_FUSE=$fuse python bench.py --width $1 --height $2 --no-e2e --no-cpu-baseline --no-pcie --steps 200 --frames-in-flight $fif 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1x$2 fuse=$fuse in flight $fif:', d['value'], d['ms_per_step'])"
  done; done
done | tee $O/r04_path_choice_settled.txt
