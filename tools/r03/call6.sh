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
EXTRA="--gab 1 --epf 2"; run "8K gab+epf2 two-phase" JXLHIP_FUSE=0; run "8K gab+epf2 fused pc0" JXLHIP_FUSE_GAB_EPF2=1 JXLHIP_FUSED_PC=0;  run "8K gab+epf2 fused pc1" JXLHIP_FUSE_GAB_EPF2=1
EXTRA="--gab 0 --epf 2"; run "8K epf2 two-phase" JXLHIP_FUSE=0; run "8K epf2 pc0" JXLHIP_FUSED_PC=0; run "8K epf2 pc1" JXLHIP_FUSED_PC=1
EXTRA="--gab 0 --epf 1"; run "8K epf1 pc0" JXLHIP_FUSED_PC=0; run "8K epf1 pc1" JXLHIP_FUSED_PC=1
EXTRA="--gab 1 --epf 0"; run "8K gab pc0" JXLHIP_FUSED_PC=0; run "8K gab pc1" JXLHIP_FUSED_PC=1
EXTRA="--config c5"; run "c5" A=1
EXTRA="--config c1"; run "c1" A=1
EXTRA="--gab 1 --epf 3"; run "8K epf3" A=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "fused" 2>&1 | tail -3
