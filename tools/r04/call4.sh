#!/bin/bash
# round 4, GPU call 4: frames in flight; e2e with the seam's own clock
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python tools/r04/inflight.py 300 2>&1 | grep -v amdgpu.ids | tee $O/r04_inflight.txt
timeout 900 python bench.py --steps 100 --no-pcie > $O/r04_call4_bench.json 2> $O/r04_call4_bench.err
python - <<PY
import json
d=json.loads(open("$O/r04_call4_bench.json").readline())
print(d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
print(json.dumps(d.get("e2e"), indent=1))
PY
