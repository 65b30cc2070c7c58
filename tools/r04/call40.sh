#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | tee $O/r04_final_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/r04_final_tests.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench --steps 20:', d['value'], d['ms_per_step'], 'e2e', d['e2e']['codestream_8k_rgb']['value'], d['e2e']['codestream_8k_rgb']['ms_per_file'], 'cpu', d['cpu_baseline']['value'])" | tee $O/r04_final_bench20.txt
