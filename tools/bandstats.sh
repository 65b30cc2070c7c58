#!/bin/bash
# GPU box: per-frame kernel time sums in banded decode_frame mode. usage: tools/bandstats.sh <band_rows>
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p
JXLHIP_BAND_ROWS=$1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > /tmp/log 2>&1
echo "== band_rows=$1"; grep -o "\"value\": [0-9.]*" /tmp/log
f=$(find /tmp/p -name "*kernel_stats.csv" | head -1)
python - $f <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "jxlhip" in n and "dequant" not in n:
        print("%-40s calls %5s total/frame %8.1f us  avg %8.1f" % (n.split("(")[0].replace("void jxlhip::","")[:40] or "k_filters*", r["Calls"], float(r["TotalDurationNs"])/1e3/43, float(r["AverageNs"])/1e3))
PY
