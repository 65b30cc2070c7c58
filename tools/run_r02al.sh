#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py tests/test_multi.py -m gpu -x -q > $O/r02al_tests.txt 2>&1; tail -3 $O/r02al_tests.txt
rm -f $O/r02al.txt
for sz in "1920 1080" "3840 2160" "7680 4320"; do
  w=${sz% *}; h=${sz#* }
  JXLHIP_FUSE=0 python bench.py --config c3 --width $w --height $h --no-pcie --no-cpu-baseline --steps 50 --warmup 5 > /tmp/b.log 2>&1
  echo "${w}x${h} two-phase $(grep -o '"value": [0-9.]*' /tmp/b.log) $(grep -o 'kernel_ms.: {[^}]*}' /tmp/b.log)" >> $O/r02al.txt
done
python bench.py --config c1 --no-pcie --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | grep -o '"value": [0-9.]*\|kernel_ms.: {[^}]*}' | tr '\n' ' ' >> $O/r02al.txt
cat $O/r02al.txt
