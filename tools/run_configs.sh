#!/bin/bash
# GPU box: one bench line per BASELINE config (N=1) + the real-content mix; tag = $1
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; tag=${1:-r02}
rm -f $O/${tag}_configs.jsonl
for c in c1 c2 c3 c4 c5; do
  extra=""; [ $c != c3 ] && [ $c != c1 ] && extra="--no-cpu-baseline"
  timeout 600 python bench.py --config $c --steps 30 --warmup 5 $extra 2>/dev/null | grep '^{' >> $O/${tag}_configs.jsonl
done
timeout 600 python bench.py --mix real4k --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' >> $O/${tag}_configs.jsonl
python - <<PY
import json
for l in open("$O/${tag}_configs.jsonl"):
    d=json.loads(l); r=d["roofline"]
    print(d["config"]["workload"][:60].ljust(60), "value %9.1f  ms %.4f  frac %.3f step %.3f kern %.3f" % (d["value"], d["ms_per_step"], r["frac"], r["frac_step"], r["frac_kernel"]), d["config"]["kernel_ms"], (d.get("pcie_inclusive") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
PY
