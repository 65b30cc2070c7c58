#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "three_epf_iterations" 2>&1 | tail -5 | tee $O/r04_call45_tests.txt
for fuse in 0 1; do JXLHIP_FUSE=$fuse python bench.py --epf 3 --no-e2e --no-cpu-baseline --no-pcie --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('epf3 JXLHIP_FUSE=$fuse:', d['value'], d['ms_per_step'], (d.get('one_frame_in_flight') or {}).get('value'), d['config']['kernel_ms'])"; done | tee $O/r04_epf3_fused2.txt
