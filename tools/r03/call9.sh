#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "producer_consumer" 2>&1 | tail -8
B="python bench.py --no-cpu-baseline --no-pcie --steps 100 --warmup 10"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/tmp.json 2>$O/tmp.err; python - <<PY
import json
try:
    d=json.load(open("$O/tmp.json")); print("$tag", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
except Exception as e: print("$tag FAILED", open("$O/tmp.err").read()[-400:])
PY
}
run "pc0" JXLHIP_FUSED_PC=0
run "pc1" JXLHIP_FUSED_PC=1
run "pc2" JXLHIP_FUSED_PC=2
for rh in 160 200 240 312 392 480; do run "pc2 rh=$rh" JXLHIP_FUSED_PC=2 JXLHIP_FUSED_PC_RH=$rh; done
EXTRA="--config c4"; run "c4 pc1" JXLHIP_FUSED_PC=1; run "c4 pc2" JXLHIP_FUSED_PC=2
EXTRA="--mix real4k"; run "real8k pc1" JXLHIP_FUSED_PC=1; run "real8k pc2" JXLHIP_FUSED_PC=2
EXTRA="--width 5120 --height 2880"; run "5k pc1" JXLHIP_FUSED_PC=1; run "5k pc2" JXLHIP_FUSED_PC=2
EXTRA="--gab 1 --epf 2"; run "gab+epf2 pc1" JXLHIP_FUSED_PC=1; run "gab+epf2 pc2" JXLHIP_FUSED_PC=2
