#!/bin/bash
# round 5: A/B of experiment builds on one box.  usage: tools/r05/ab.sh <tag> "<pytest -k expr or empty>" <variant tags...>
# (variant "" = the product build); every line = bench.py one frame in flight, 200 steps
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
tag=$1; kexpr=$2; shift 2
V=$R/libjxl_amd/csrc/variants
if [ -n "$kexpr" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "$kexpr" 2>&1 | tail -15 > $O/${tag}_tests.txt
  cat $O/${tag}_tests.txt
fi
q() { bash tools/quick.sh "$1" --no-pcie --frames-in-flight 1 --steps 200 --warmup 20 "${@:2}"; }
{
for rep in 1 2; do
  q ""
  for v in "$@"; do q "JXLHIP_SO=$V/libjxl_hip_$v.so"; done
done
} 2>&1 | tee $O/${tag}_bench.txt
