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
V=$R/libjxl_amd/csrc/variants
run "pc0" JXLHIP_FUSED_PC=0
run "pc1" JXLHIP_FUSED_PC=1
run "onebuf v1 (8/CU, 4 waves/SIMD)" JXLHIP_SO=$V/libjxl_hip_pc_onebuf.so
for rh in 104 128 152 176 200; do run "onebuf v1 rh=$rh" JXLHIP_SO=$V/libjxl_hip_pc_onebuf.so JXLHIP_FUSED_PC_RH=$rh; done
run "onebuf v3" JXLHIP_SO=$V/libjxl_hip_pc_onebuf_v3.so
for rh in 128 152 200; do run "onebuf v3 rh=$rh" JXLHIP_SO=$V/libjxl_hip_pc_onebuf_v3.so JXLHIP_FUSED_PC_RH=$rh; done
