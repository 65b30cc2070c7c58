#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -m gpu -x -q -k "fused or config or pipeline" > $O/r02r_tests.txt 2>&1; tail -4 $O/r02r_tests.txt
rm -f $O/r02r.txt
one() { e="$1"; shift
  v=$(env $e python bench.py "$@" --no-pcie --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep -o '"value": [0-9.]*\|kernel_ms.: {[^}]*}' | tr '\n' ' ')
  echo "$e | $* | $v" >> $O/r02r.txt; }
for rh in 32 40 48 64 80; do one "JXLHIP_FUSE=1 JXLHIP_FUSED_RH=$rh" --config c2 --gab 1 --epf 1; done
one "JXLHIP_FUSE=0" --config c2 --gab 1 --epf 1
one "JXLHIP_FUSE=1" --config c2
one "JXLHIP_FUSE=0" --config c2
for rh in 72 104 136; do one "JXLHIP_FUSED_RH=$rh" --config c3; done
one "A=1" --config c3
one "A=1" --config c3 --gab 0 --epf 0
one "A=1" --config c5
one "JXLHIP_FUSE=0" --config c5
one "JXLHIP_FUSE=1" --config c1
one "JXLHIP_FUSE=0" --config c1
cat $O/r02r.txt
