#!/bin/bash
# round 5: where a djxl repetition goes -- float frame (--disable_output) and 8-bit PPM, three tools
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
E=$R/oracle/_ref; F=$R/tests/data/e2e_8k_d1.jxl; S8=$R/tests/data/e2e_8k_d1_srgb8.jxl
export LD_LIBRARY_PATH=$E:$R/libjxl_amd/csrc:$LD_LIBRARY_PATH JXLHIP_SEAM_VERBOSE=1
{
for th in 16 64; do
for tool in djxl_hip djxl_ref_v8 djxl_ref; do
  echo "== $tool float --disable_output threads $th"; $E/$tool $F --disable_output --num_reps 10 --num_threads $th 2>&1 | grep "MP/s\|jxlhip seam: frame" | tail -2
  echo "== $tool 8-bit ppm threads $th"; $E/$tool $S8 /tmp/o_$tool.ppm --num_reps 10 --num_threads $th 2>&1 | grep "MP/s\|jxlhip seam" | tail -2
done; done
cmp /tmp/o_djxl_ref.ppm /tmp/o_djxl_ref_v8.ppm && echo "ref == ref_v8 (ppm bytes)"
python - <<'PY'
import numpy as np
def rd(p):
    b=open(p,'rb').read(); h=b.split(b"\n",3); w,hh=map(int,h[1].split()); return np.frombuffer(h[3],np.uint8).reshape(hh,w,3)
a=rd('/tmp/o_djxl_ref.ppm').astype(int); b=rd('/tmp/o_djxl_hip.ppm').astype(int); c=rd('/tmp/o_djxl_ref_v8.ppm').astype(int)
print("hip vs ref: max", abs(a-b).max(), "differing", (a!=b).mean(), " v8 vs ref: max", abs(a-c).max(), "differing", (a!=c).mean())
PY
} 2>&1 | tee $O/djxl_e2e.txt
