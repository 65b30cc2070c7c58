#!/bin/bash
# GPU box: wave-instructions per launch of every kernel of the bench step (SQ_INSTS_VALU / SQ_INSTS_SALU and the wave /
# busy-cycle counters beside them), one rocprofv3 --pmc pass of the bench command -> gpurun_out/pmc_valu_<tag>.json,
# the source of bench.py's `roofline_valu` (copied to profiles/pmc_valu.json by the builder, like pmc_traffic.json).
# usage: tools/pmc_valu.sh <tag> [commit] [bench args]
tag=${1:-run}; commit=${2:-unknown}; shift; shift
export JXLHIP_BENCH_NO_GRAPH=1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_valu
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d /tmp/pmc_valu -- \
  python $R/bench.py --no-cpu-baseline --no-pcie --no-e2e --frames-in-flight 1 --steps 5 --warmup 2 "$@" > $O/pmc_valu_$tag.log 2>&1
f=$(find /tmp/pmc_valu -name "*counter_collection.csv" | head -1)
python - "$f" "$commit" > $O/pmc_valu_$tag.json <<'PY'
import csv, json, sys
from collections import defaultdict
slots = (("k_fused", "filters"), ("k_filters", "filters"), ("k_epf0", "epf0"), ("k_transform_mfma32", "blocks_mfma32"), ("k_transform_mfma16", "blocks_mfma16"),
         ("k_transform_8", "blocks_8x8"), ("k_transform_r16", "blocks_r16"), ("k_transform_r32", "blocks_r32"), ("k_transform_r", "blocks_r"),
         ("k_transform_a", "blocks_a"), ("k_large", "blocks_large"), ("k_prepare", "prepare"))
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "jxlhip" not in n or "k_dequant_tables" in n:
        continue
    slot = next((s for k, s in slots if k in n), None)
    if slot:
        acc[slot][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {s: {c: round(sum(v) / len(v)) for c, v in d.items()} for s, d in acc.items()}
for s, d in acc.items():
    out[s]["launches_seen"] = max(len(v) for v in d.values())
out["_commit"] = sys.argv[2]
out["_what"] = "average per launch over the launches of the run (rocprofv3 --pmc, one pass); wave-instruction counts (SQ_INSTS_*)"
print(json.dumps(out, indent=1))
PY
cat $O/pmc_valu_$tag.json
