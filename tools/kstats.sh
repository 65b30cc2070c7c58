#!/bin/bash
# GPU box: per-kernel durations of one bench.py configuration (rocprofv3 --kernel-trace --stats)
# usage: tools/kstats.sh "<ENV=.. ENV=..>" [bench args]
envs="$1"; shift
export JXLHIP_BENCH_NO_GRAPH=1  # (profiling / experiment runs: no hipGraph side measurement)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p
env $envs timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python $R/bench.py --no-cpu-baseline --frames-in-flight 1 "$@" > /tmp/log 2>&1
echo "== env=[$envs] args=[$@]"
grep -o "\"value\": [0-9.]*\|kernel_ms.: {[^}]*}" /tmp/log || tail -5 /tmp/log
f=$(find /tmp/p -name "*kernel_stats.csv" | head -1)
python - $f <<PY
import csv,sys
tot=0
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "jxlhip" in n:
        print("%-64s calls %4s avg %8.1f us  min %8.1f" % (n.replace("(anonymous namespace)::","").split("(")[0].replace("void jxlhip::","")[:64], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
