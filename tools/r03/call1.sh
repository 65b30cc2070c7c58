#!/bin/bash
# Round 3, GPU call 1: the GPU suite at HEAD (djxl on the HIP back-end, conformance mini-corpus, multi no-sync,
# producer/consumer fused kernel parity), then k_fused vs k_fused_pc on the bench workload.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -150 > $O/r03_pytest_gpu_1.log
tail -40 $O/r03_pytest_gpu_1.log
B="python bench.py --no-cpu-baseline --no-pcie --steps 100 --warmup 10"
for pc in 0 1; do
  JXLHIP_FUSED_PC=$pc timeout 300 $B > $O/r03_bench_c3_pc$pc.json 2> $O/r03_bench_c3_pc$pc.err
  python - <<PY
import json; d=json.load(open("$O/r03_bench_c3_pc$pc.json")); print("c3 pc=$pc", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
for rh in 64 104 136 200 280 440; do
  JXLHIP_FUSED_PC=1 JXLHIP_FUSED_PC_RH=$rh timeout 300 $B > $O/r03_bench_c3_pc1_rh$rh.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("$O/r03_bench_c3_pc1_rh$rh.json")); print("c3 pc=1 rh=$rh", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
for cfg in c2 c4; do for pc in 0 1; do
  JXLHIP_FUSED_PC=$pc timeout 300 $B --config $cfg > $O/r03_bench_${cfg}_pc$pc.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("$O/r03_bench_${cfg}_pc$pc.json")); print("$cfg pc=$pc", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
PY
done; done
for pc in 0 1; do
  JXLHIP_FUSED_PC=$pc timeout 300 $B --mix real4k > $O/r03_bench_real8k_pc$pc.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("$O/r03_bench_real8k_pc$pc.json")); print("real8k pc=$pc", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
PY
done
# 4K two-phase vs fused-PC forced (is the 12 Mpx threshold still right?)
for w in "3840 2160" "5120 2880"; do set -- $w; for mode in "JXLHIP_FUSE=0" "JXLHIP_FUSE=1 JXLHIP_FUSED_PC=0" "JXLHIP_FUSE=1 JXLHIP_FUSED_PC=1"; do
  env $mode timeout 300 $B --width $1 --height $2 > $O/tmp.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("$O/tmp.json")); print("$1x$2 $mode", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
PY
done; done
# the transform kernel without spills (2 waves / SIMD)
JXLHIP_SO=$R/libjxl_amd/csrc/variants/libjxl_hip_rw2.so JXLHIP_FUSED_PC=1 timeout 300 $B > $O/r03_bench_c3_rw2.json 2>/dev/null
python - <<PY
import json; d=json.load(open("$O/r03_bench_c3_rw2.json")); print("c3 pc=1 R_WAVES=2", d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pc; JXLHIP_FUSED_PC=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pc -- python $R/bench.py --no-cpu-baseline --no-pcie --steps 50 > $O/r03_prof_pc.log 2>&1
f=$(find /tmp/prof_pc -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r03_c3_pc_kernel_stats.csv && head -8 $O/r03_c3_pc_kernel_stats.csv | cut -c1-200
