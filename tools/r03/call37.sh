#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python - <<'PY'
import numpy as np
for n in ("4k", "8k"):
    d = np.load(f"oracle/_ref/real_{n}_rgba.npz"); open(f"/tmp/rgba{n}.jxl", "wb").write(d["codestream"].tobytes())
PY
for f in 4k 8k; do for ext in pam ppm; do for thr in 16 64; do
  echo -n "whole file rgba$f -> .$ext, $thr workers: "; timeout 200 python tools/djxl_hip.py /tmp/rgba$f.jxl /tmp/o.$ext --threads $thr --reps 5 2>&1 | grep -i "mp/s\|mpx\|error" | tail -1
done; done; done
for f in 4k 8k; do
  for tool in djxl_ref djxl_hip; do echo -n "$tool rgba$f -> pam: "; JXLHIP_SEAM_VERBOSE=1 timeout 300 oracle/_ref/$tool /tmp/rgba$f.jxl /tmp/o.pam --num_reps 3 2>&1 | grep "MP/s\|seam: frame" | tr '\n' ' ' | cut -c1-400; echo; done
done
