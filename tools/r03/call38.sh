#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 1200 python -m pytest tests/test_extra_channels.py tests/test_djxl.py tests/test_seam.py tests/test_codestream.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-300
python - <<'PY'
import numpy as np
for n in ("4k", "8k"):
    d = np.load(f"oracle/_ref/real_{n}_rgba.npz"); open(f"/tmp/rgba{n}.jxl", "wb").write(d["codestream"].tobytes())
PY
for f in 4k 8k; do for thr in 16 64; do
  echo -n "whole file rgba$f -> .pam, $thr workers: "; timeout 200 python tools/djxl_hip.py /tmp/rgba$f.jxl /tmp/o.pam --threads $thr --reps 5 2>&1 | grep -i "mp/s\|mpx\|error" | tail -1 | cut -c1-80
done; done
for f in 4k 8k; do
  for tool in djxl_hip; do echo -n "$tool rgba$f -> pam: "; JXLHIP_SEAM_VERBOSE=1 timeout 300 oracle/_ref/$tool /tmp/rgba$f.jxl /tmp/o.pam --num_reps 3 2>&1 | grep "MP/s\|seam: frame" | tail -2 | tr '\n' ' ' | cut -c100-500; echo; done
done
