#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
for v in "A=1" "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_noatomic.so"; do
  bash tools/kstats.sh "$v" --config c3 --no-pcie --steps 30 --warmup 5 2>&1 | grep "env=\|k_prepare" | cut -c1-130
  bash tools/kstats.sh "$v" --config c4 --no-pcie --steps 10 --warmup 3 2>&1 | grep "k_prepare" | cut -c1-130
done
