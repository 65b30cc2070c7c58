#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
for role in -1; do JXLHIP_FUSED_PC0_ROLE=$role JXLHIP_FUSE=1 python bench.py --epf 3 --no-e2e --no-cpu-baseline --no-pcie --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('epf3 fused, role shift $role:', d['value'], d['ms_per_step'], (d.get('one_frame_in_flight') or {}).get('value'), d['config']['kernel_ms'])"; done | tee $O/r04_epf3_roles.txt
