#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "fused or tile or stripe" 2>&1 | tail -3
one() { env $1 python bench.py --no-cpu-baseline --no-pcie --steps 300 --warmup 30 --frames-in-flight $2 "${@:3}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-40s in flight $2: %8.1f Gpx/s  %s' % ('$1'[-40:], d['value']/1e3, d['config']['kernel_ms']))"; }
for rep in 1 2; do
  one "X=shipped_prio3" 1; one "X=shipped_prio3" 3
  one "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_prodprio1.so" 1; one "JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_prodprio1.so" 3
done
one "X=shipped_prio3" 3 --mix real4k; one "X=shipped_prio3" 3 --config c2; one "X=shipped_prio3" 3 --config c4; one "X=shipped_prio3" 3 --epf 2; one "X=shipped_prio3" 3 --epf 3
