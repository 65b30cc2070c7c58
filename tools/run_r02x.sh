#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02x.txt
for v in pk; do
  JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$v.so timeout 600 python -m pytest tests/test_gpu_vs_reference.py -m gpu -x -q -k "config2_8k" 2>&1 | tail -2 >> $O/r02x.txt
done
for v in base pk base pk; do
  bash tools/kstats.sh "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$v.so" --config c3 --no-pcie --steps 30 --warmup 5 2>&1 | grep "env=\|value\|k_fused\|k_transform_r" >> $O/r02x.txt
done
cat $O/r02x.txt
