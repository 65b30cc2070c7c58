#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
for m in 0 1; do
  bash tools/kstats.sh "JXLHIP_MFMA=$m" --config c2 --mix real4k --gab 1 --epf 1 --no-pcie --steps 20 --warmup 5 > $O/r02p_real4k_mfma$m.txt 2>&1; cat $O/r02p_real4k_mfma$m.txt
done
bash tools/kstats.sh "JXLHIP_FUSE=0" --config c2 --mix real4k --gab 1 --epf 1 --no-pcie --steps 20 --warmup 5 > $O/r02p_real4k_two.txt 2>&1; cat $O/r02p_real4k_two.txt
bash tools/kstats.sh "A=1" --config c3 --mix real4k --no-pcie --steps 20 --warmup 5 > $O/r02p_real8k.txt 2>&1; cat $O/r02p_real8k.txt
bash tools/kstats.sh "A=1" --config c3 --mix 18:1 --no-pcie --steps 20 --warmup 5 > $O/r02p_all64.txt 2>&1; cat $O/r02p_all64.txt
