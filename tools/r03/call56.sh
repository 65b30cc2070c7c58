#!/bin/bash
# fused (k_fused_pc) vs two-phase (k_transform_* + k_filters_fast) by strategy mix at 8K and 4K
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
one() { python bench.py --no-cpu-baseline --no-pcie --steps 200 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%8.1f Gpx/s  %s' % (d['value']/1e3, {k: round(v,4) for k,v in d['config']['kernel_ms'].items()}))"; }
for f in 1 0 1; do
  export JXLHIP_FUSE=$f
  echo "== JXLHIP_FUSE=$f"; echo -n "c3 (d1 mix) "; one; echo -n "real8k      "; one --mix real4k; echo -n "real 4K     "; one --mix real4k --width 3840 --height 2160
done
