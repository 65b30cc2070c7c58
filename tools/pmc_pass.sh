#!/bin/bash
# Usage (GPU box): tools/pmc_pass.sh "<counter list>" [bench args]  -> per-kernel averages
ctrs="$1"; shift
export JXLHIP_BENCH_NO_GRAPH=1  # (profiling / experiment runs: no hipGraph side measurement)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcx
timeout ${PMC_TIMEOUT:-600} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmcx -- python $R/bench.py --no-cpu-baseline --frames-in-flight 1 --steps 3 --warmup 1 "$@" > /tmp/pmcx.log 2>&1
f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "jxlhip" not in n: continue
    n = n.replace("(anonymous namespace)::","").split("(")[0].replace("void jxlhip::","")[:34]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("kernel".ljust(36) + " ".join(c[-18:].rjust(19) for c in names))
for k, d in sorted(acc.items()):
    print(k.ljust(36) + " ".join(("%.4g" % (sum(d[c])/len(d[c])) if c in d else "-").rjust(19) for c in names))
PY
