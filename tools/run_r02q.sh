#!/bin/bash
# rows-per-wave sweep of the fused kernel (4K, 8K; d1 and real mixes) and of the fast filter at 4K
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02q.txt
one() { # env, args...
  e="$1"; shift
  v=$(env $e python bench.py "$@" --no-pcie --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep -o '"value": [0-9.]*\|kernel_ms.: {[^}]*}' | tr '\n' ' ')
  echo "$e | $* | $v" >> $O/r02q.txt
}
for rh in 24 32 40 48 64 80 96 136; do
  one "JXLHIP_FUSED_RH=$rh" --config c2 --gab 1 --epf 1
done
for rh in 24 48 64 96; do
  one "JXLHIP_FUSED_RH=$rh" --config c2 --gab 1 --epf 1 --mix real4k
done
one "JXLHIP_FUSE=0" --config c2 --gab 1 --epf 1
for rh in 56 72 104 136 184 272; do
  one "JXLHIP_FUSED_RH=$rh" --config c3
done
one "JXLHIP_FUSE=0" --config c3 --mix real4k
for rh in 104 184; do
  one "JXLHIP_FUSED_RH=$rh" --config c3 --mix real4k
done
one "A=1" --config c1
one "JXLHIP_FUSE=0" --config c1
for rh in 16 32 64; do one "JXLHIP_FUSED_RH=$rh" --config c1; done
cat $O/r02q.txt
