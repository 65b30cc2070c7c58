#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
one() { L="${*:3}"; env $1 python bench.py --no-cpu-baseline --no-pcie --steps 300 --warmup 30 --frames-in-flight $2 "${@:3}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-24s in flight $2 $L: %8.1f Gpx/s  %s' % ('$1', d['value']/1e3, d['config']['kernel_ms']))"; }
one JXLHIP_FUSE=0 3 --mix real4k; one JXLHIP_FUSE=1 3 --mix real4k
one X=1 3 --width 3840 --height 2160; one JXLHIP_FUSE=1 3 --width 3840 --height 2160
one X=1 3 --width 3840 --height 2160 --mix real4k; one JXLHIP_FUSE=1 3 --width 3840 --height 2160 --mix real4k
one X=1 3 --width 1920 --height 1080; one JXLHIP_FUSE=1 3 --width 1920 --height 1080
