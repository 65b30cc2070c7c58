#!/bin/bash
# call58 again for the small frames, 1000 steps, the builds interleaved twice: dynamic (default build), static, hybrid
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
one() { python bench.py --no-cpu-baseline --no-pcie --steps 1000 --warmup 50 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f Gpx/s  blocks %.4f' % (d['value']/1e3, d['config']['kernel_ms'].get('blocks',0)))"; }
export JXLHIP_FUSE=0
for rep in 1 2; do for so in default d8static d8hybrid; do
  if [ $so = default ]; then unset JXLHIP_SO; else export JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$so.so; fi
  echo -n "$so 4K d1   "; one --width 3840 --height 2160; echo -n "$so 4K real "; one --mix real4k --width 3840 --height 2160
done; done
