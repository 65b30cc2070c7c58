#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "mfma" 2>&1 | tail -5
for m in 0 1; do
  bash tools/quick.sh "JXLHIP_MFMA=$m" --config c3 --mix 4:1 --gab 0 --epf 0 --steps 50 --no-pcie
  bash tools/quick.sh "JXLHIP_MFMA=$m" --config c3 --mix 4:1 --steps 50 --no-pcie
  bash tools/quick.sh "JXLHIP_MFMA=$m" --config c3 --mix 4:1,5:1 --steps 50 --no-pcie
  bash tools/quick.sh "JXLHIP_MFMA=$m" --config c3 --steps 50 --no-pcie
  bash tools/quick.sh "JXLHIP_MFMA=$m" --config c2 --mix 4:1 --steps 50 --no-pcie
done
