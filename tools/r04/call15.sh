#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R
timeout 900 python bench.py --steps 50 --no-pcie > $O/r04_call15_bench.json 2> $O/r04_call15_bench.err; tail -3 $O/r04_call15_bench.err
python - <<PY
import json
d=json.loads(open("$O/r04_call15_bench.json").readline())
print(d["value"], d["one_frame_in_flight"]["value"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("best_rep"))
print(json.dumps(d["e2e"].get("codestream_8k_rgb_files_in_flight"), indent=1))
print(d["e2e"]["codestream_8k_rgb"]["value"], d["e2e"]["djxl_hip"]["value"], d["e2e"]["djxl_ref"]["value"])
PY
