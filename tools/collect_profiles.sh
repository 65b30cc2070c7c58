#!/bin/bash
# build container: copies what tools/regen_profiles.sh <tag> left in gpurun_out/ into profiles/ (tracked)
tag=${1:-r03}; cd $(dirname $0)/..
for cfg in c1 c2 c3 c4 c5 real8k c3_epf3; do
  for f in bench.json kernel_stats.csv; do [ -s gpurun_out/${tag}_${cfg}_$f ] && cp gpurun_out/${tag}_${cfg}_$f profiles/${tag}_${cfg}_$f; done
  [ -s gpurun_out/pmc_traffic_${tag}_$cfg.json ] && cp gpurun_out/pmc_traffic_${tag}_$cfg.json profiles/${tag}_${cfg}_pmc_traffic.json
done
for f in c3_bench_full.json c5_mfma_counters.txt c3_mfma_counters.txt commit.txt; do [ -s gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f; done
# the file bench.py replays for roofline.traffic: the c3 passes, stamped with the commit they were taken at
python3 - <<PY
import json
d = json.load(open("profiles/${tag}_c3_pmc_traffic.json"))
d["_commit"] = open("profiles/${tag}_commit.txt").read().strip()
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
print({k: v for k, v in d.items() if not k.startswith("_") and not isinstance(v, dict)})
PY
