#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 900 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err ) 2>&1 | tail -3
python - <<PY
import json
d=json.load(open("$O/r03_bench_default.json"))
print(json.dumps({k: d[k] for k in ("value","ms_per_step","roofline","pcie_inclusive","cpu_baseline")}, indent=1)[:3000])
PY
tail -5 $O/r03_bench_default.err
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "stripes_through" 2>&1 | tail -3
