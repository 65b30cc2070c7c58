#!/bin/bash
# round 4, GPU call 3: the default bench line with the new e2e block (whole-file rates), the launch timeline of the c3 step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python bench.py > $O/r04_call3_bench_full.json 2> $O/r04_call3_bench_full.err
python - <<PY
import json
d=json.loads(open("$O/r04_call3_bench_full.json").readline())
print(d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
print(json.dumps(d.get("e2e"), indent=1))
print(json.dumps(d.get("cpu_baseline"), indent=1)[:600])
PY
tail -5 $O/r04_call3_bench_full.err
timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "tile" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --no-cpu-baseline --no-pcie --steps 60 --warmup 5 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1); echo "trace: $f"; tail -2 /tmp/tl.log | cut -c1-300; head -c 600 $f; python $R/tools/timeline.py $f
