#!/bin/bash
# Round 3, GPU call 2: which wave bounds k_fused_pc?  Ablation builds (garbage pixels, timing only).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_djxl.py -m gpu -q --tb=short 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-pcie --steps 100 --warmup 10"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/tmp.json 2>$O/tmp.err; python - <<PY
import json
try:
    d=json.load(open("$O/tmp.json")); print("$tag", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
except Exception as e: print("$tag FAILED", open("$O/tmp.err").read()[-400:])
PY
}
V=$R/libjxl_amd/csrc/variants
run "pc0" JXLHIP_FUSED_PC=0
run "pc1" JXLHIP_FUSED_PC=1
run "pc1 nofill" JXLHIP_FUSED_PC=1 JXLHIP_SO=$V/libjxl_hip_pc_nofill.so
run "pc1 nomarch" JXLHIP_FUSED_PC=1 JXLHIP_SO=$V/libjxl_hip_pc_nomarch.so
run "pc1 nostore" JXLHIP_FUSED_PC=1 JXLHIP_SO=$V/libjxl_hip_pc_nostore.so
for rh in 104 200 280; do
run "pc1 nofill rh=$rh" JXLHIP_FUSED_PC=1 JXLHIP_FUSED_PC_RH=$rh JXLHIP_SO=$V/libjxl_hip_pc_nofill.so
run "pc1 nomarch rh=$rh" JXLHIP_FUSED_PC=1 JXLHIP_FUSED_PC_RH=$rh JXLHIP_SO=$V/libjxl_hip_pc_nomarch.so
done
