#!/bin/bash
# round 4, GPU call 5: interior-first stripes (jxlhip_decode_filters_rows), multi-context, 8K through djxl_hip; the default
# bench line with frames in flight
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_multi.py -q -k "stripe or multi" 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_djxl.py -q -m gpu -k "8k or mini_corpus or 4k" 2>&1 | tail -8
timeout 900 python bench.py --steps 100 --no-pcie --no-e2e --no-cpu-baseline > $O/r04_call5_bench.json 2> $O/r04_call5_bench.err; tail -2 $O/r04_call5_bench.err
python - <<PY
import json
d=json.loads(open("$O/r04_call5_bench.json").readline())
print(d["value"], d["ms_per_step"], d["config"]["kernel_ms"], d.get("one_frame_in_flight"), d["roofline"]["frac"])
PY
for f in 1 2 3 4; do python bench.py --steps 200 --no-pcie --no-e2e --no-cpu-baseline --frames-in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('in flight $f', d['value'], d['ms_per_step'])"; done
for cfg in c2 c1 c5 c4; do python bench.py --config $cfg --steps 100 --no-pcie --no-e2e --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg', d['value'], d['ms_per_step'], d.get('one_frame_in_flight',{}).get('value'))"; done
