#!/bin/bash
# k_transform_r, two-phase: only the DCT8 workgroups that have work alternate with the persistent ones (default build)
# against every workgroup of the worst-case bound alternating (variant d8static = the behaviour until this commit)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 260 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 > $O/full_gpu.log; tail -2 $O/full_gpu.log
one() { python bench.py --no-cpu-baseline --no-pcie --steps 200 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f Gpx/s  %s' % (d['value']/1e3, {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}))"; }
for so in default d8static; do
  if [ $so = default ]; then unset JXLHIP_SO; else export JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$so.so; fi
  echo "== $so"
  export JXLHIP_FUSE=0
  echo -n "8K d1   two-phase "; one; echo -n "8K real two-phase "; one --mix real4k; echo -n "4K real two-phase "; one --mix real4k --width 3840 --height 2160
  echo -n "4K d1   two-phase "; one --width 3840 --height 2160; echo -n "1080p d1          "; one --width 1920 --height 1080; echo -n "1080p real        "; one --mix real4k --width 1920 --height 1080
  unset JXLHIP_FUSE
done
