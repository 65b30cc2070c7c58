#!/bin/bash
# k_transform_r<short> occupancy / spill trade-off: JXLHIP_R_WAVES = 3 (shipped: 168 VGPRs, spills), 2 (256 VGPRs), 4 (128)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
one() { python bench.py --no-cpu-baseline --no-pcie --steps 200 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f Gpx/s  blocks %.4f fused %.4f ms' % (d['value']/1e3, d['config']['kernel_ms'].get('blocks',0), d['config']['kernel_ms'].get('fused',0)))"; }
for so in default rw2 rw4 default; do
  if [ $so = default ]; then unset JXLHIP_SO; else export JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_$so.so; fi
  echo "== $so"; echo -n "c3     "; one; echo -n "real8k "; one --mix real4k; echo -n "c2(4K) "; one --config c2
done
