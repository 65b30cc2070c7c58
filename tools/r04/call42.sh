#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py > $O/r04_c3_bench_full.json 2>/dev/null; tail -c 400 $O/r04_c3_bench_full.json
python bench.py --steps 20 --warmup 5 > $O/r04_bench_steps20.json 2>/dev/null
python bench.py --mix real4k --no-e2e --no-cpu-baseline --no-pcie --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('real mix:', d['value'], d['ms_per_step'], (d.get('one_frame_in_flight') or {}).get('value'))" | tee $O/r04_real_mix_settled.txt
python bench.py --epf 3 --no-e2e --no-cpu-baseline --no-pcie --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('epf3:', d['value'], d['ms_per_step'], (d.get('one_frame_in_flight') or {}).get('value'))" | tee -a $O/r04_real_mix_settled.txt
for c in c1 c2 c4; do python bench.py --config $c --no-e2e --no-cpu-baseline --no-pcie --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$c:', d['value'], d['ms_per_step'], (d.get('one_frame_in_flight') or {}).get('value'))"; done | tee -a $O/r04_real_mix_settled.txt
