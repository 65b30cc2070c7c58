#!/bin/bash
# ablations of k_fused (variant builds): what do the plane DMA and the in-wave DCT8 decode cost?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
rm -f $O/r02s.txt
for v in lean nodma nodec neither; do
  bash tools/kstats.sh "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$v.so" --config c3 --no-pcie --steps 20 --warmup 5 >> $O/r02s.txt 2>&1
done
bash tools/kstats.sh "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_lean.so JXLHIP_DEBUG=4" --config c3 --no-pcie --steps 20 --warmup 5 >> $O/r02s.txt 2>&1
cat $O/r02s.txt
