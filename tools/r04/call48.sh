#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | tee $O/r04_final_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/r04_final_tests.txt
python bench.py --epf 3 --no-e2e --no-cpu-baseline --no-pcie --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('epf3 auto:', d['value'], d['ms_per_step'], (d.get('one_frame_in_flight') or {}).get('value'), d['config']['kernel_ms'])"
