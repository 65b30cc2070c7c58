#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
one() { L="${*:3}"; env $1 python bench.py --no-cpu-baseline --no-pcie --steps 300 --warmup 30 --frames-in-flight $2 "${@:3}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s in flight $2 $L: %8.1f Gpx/s  %s' % ('$1'[-28:], d['value']/1e3, d['config']['kernel_ms']))"; }
V="JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_rwaves2.so"
for rep in 1 2; do
one X=shipped 3; one $V 3; one X=shipped 1; one $V 1
one X=shipped 3 --mix real4k; one $V 3 --mix real4k
done
one X=shipped 3 --config c4; one $V 3 --config c4
