#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
B="python bench.py --no-cpu-baseline --no-pcie --steps 100 --warmup 10"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/tmp.json 2>$O/tmp.err; python - <<PY
import json
try:
    d=json.load(open("$O/tmp.json")); print("$tag", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
except Exception as e: print("$tag FAILED", open("$O/tmp.err").read()[-400:])
PY
}
run "pc0" JXLHIP_FUSED_PC=0
for role in -1 0 1 2 3; do run "pc1 role=$role" JXLHIP_FUSED_PC_ROLE=$role; done
for role in -1 0 1 2; do run "pc1 rh=104 role=$role" JXLHIP_FUSED_PC_RH=104 JXLHIP_FUSED_PC_ROLE=$role; done
