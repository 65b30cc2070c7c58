#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python - <<'PY'
import numpy as np
for n, f in (("8k", "oracle/_ref/real_8k_d1.npz"), ("4k", "tests/data/real_4k_d1.npz")):
    d = np.load(f); open(f"/tmp/real{n}.jxl", "wb").write(d["codestream"].tobytes())
PY
for f in 8k 4k; do for thr in 16 32 64 128; do
  echo -n "whole file real$f, $thr workers: "; timeout 120 python tools/djxl_hip.py /tmp/real$f.jxl /tmp/o.npy --threads $thr --reps 6 2>&1 | grep -i "mpx\|MP/s\|error" | tail -1
done; done
