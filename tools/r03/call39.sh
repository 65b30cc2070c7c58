#!/bin/bash
# per-class cost of phase 1 at 8K: blocks ms when the whole frame is ONE strategy (two-phase, JXLHIP_MFMA=0)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for s in 0 6 7 4 10 11 5 8 9 18 19 20 1 2 3 12 13 14 16 17; do
  echo -n "strategy $s: "; JXLHIP_FUSE=0 JXLHIP_MFMA=0 timeout 120 python bench.py --no-cpu-baseline --no-pcie --config c3 --mix $s:1 --steps 30 --warmup 3 2>&1 | grep -o "kernel_ms.: {[^}]*}" 
done
echo -n "d1 mix two-phase: "; JXLHIP_FUSE=0 timeout 120 python bench.py --no-cpu-baseline --no-pcie --config c3 --steps 30 --warmup 3 2>&1 | grep -o "kernel_ms.: {[^}]*}"
echo -n "d1 mix fused: "; timeout 120 python bench.py --no-cpu-baseline --no-pcie --config c3 --steps 30 --warmup 3 2>&1 | grep -o "kernel_ms.: {[^}]*}"
echo -n "d1 mix without DCT8 (what k_transform_r does in fused mode), two-phase: "; JXLHIP_FUSE=0 timeout 120 python bench.py --no-cpu-baseline --no-pcie --config c3 --mix "6:10,7:10,4:12,10:2.5,11:2.5,5:3,1:0.5,2:0.5,3:0.7,12:0.7,13:0.7,14:0.5,15:0.5,16:0.5,17:0.4,8:2.5,9:2.5,18:2,19:1.5,20:1.5" --steps 30 --warmup 3 2>&1 | grep -o "kernel_ms.: {[^}]*}"
