#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python - <<'PY'
import numpy as np
d = np.load("tests/data/real_4k_d1.npz"); open("/tmp/real4k.jxl", "wb").write(d["codestream"].tobytes())
PY
for tool in djxl_ref djxl_hip; do for thr in 32 128; do
  echo "== $tool --num_threads $thr"; oracle/_ref/$tool /tmp/real4k.jxl --disable_output --num_reps 10 --num_threads $thr 2>&1 | grep -i "MP/s\|error\|fail" | tail -1
done; done
JXLHIP_SEAM_VERBOSE=1 oracle/_ref/djxl_hip /tmp/real4k.jxl --disable_output --num_reps 3 --num_threads 32 2>&1 | grep -i "seam" | cut -c100-260 | tail -1
timeout 600 python -m pytest tests/test_djxl.py tests/test_seam.py -m gpu -q --tb=short 2>&1 | tail -4
