#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py > $O/r04_c3_bench_full.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > $O/r04_bench_steps20.json 2>/dev/null
python - <<'PY'
import json
for f in ('r04_c3_bench_full.json','r04_bench_steps20.json'):
    d=json.loads(open('gpurun_out/'+f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['one_frame_in_flight']['value'], d['roofline']['frac'], d['real_content_mix']['value'], d['e2e']['codestream_8k_rgb']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
