#!/usr/bin/env python3
"""GPU box: where a decode step's time goes between its launches.  usage: tools/timeline.py <kernel_trace.csv> [first kernel substring]
Reads rocprofv3 --kernel-trace output, cuts the stream of launches into steps at every `k_prepare`, and prints the
average duration of each launch and of the gap in front of it (end of the previous launch -> its start)."""
import csv
import sys
from collections import defaultdict

rows = []
rd = csv.DictReader(open(sys.argv[1]))
cols = {c.lower(): c for c in rd.fieldnames}
ks, ke, kn = cols.get("start_timestamp"), cols.get("end_timestamp"), cols.get("kernel_name")
if not (ks and ke and kn):
    sys.exit("unexpected columns: %s" % rd.fieldnames)
for r in rd:
    rows.append((int(r[ks]), int(r[ke]), r[kn]))
print("%d launches in %s" % (len(rows), sys.argv[1]))
rows.sort()
first = sys.argv[2] if len(sys.argv) > 2 else "k_prepare"
steps, cur = [], None
for s, e, n in rows:
    if first in n:
        if cur:
            steps.append(cur)
        cur = []
    if cur is not None:
        cur.append((s, e, n))
if cur:
    steps.append(cur)
if not steps:
    sys.exit("no launch named *%s*; first names: %s" % (first, [n[:50] for _, _, n in rows[:5]]))
from collections import Counter
common = Counter(len(x) for x in steps).most_common(1)[0][0]
steps = [st for st in steps if len(st) == common][5:-1]  # steady state, steps of the usual shape
acc = defaultdict(lambda: [0.0, 0.0, 0])
period = []
for i, st in enumerate(steps):
    for j, (s, e, n) in enumerate(st):
        short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void jxlhip::", "").replace("jxlhip::", "")[:60]
        prev_end = st[j - 1][1] if j else (steps[i - 1][-1][1] if i else s)
        a = acc[(j, short)]
        a[0] += (e - s) / 1e3
        a[1] += (s - prev_end) / 1e3
        a[2] += 1
    if i:
        period.append((st[0][0] - steps[i - 1][0][0]) / 1e3)
print("%d steps; step period %.1f us" % (len(steps), sum(period) / max(1, len(period))))
for (j, short), (d, g, n) in sorted(acc.items()):
    print("  %-60s  run %7.1f us   gap before %6.1f us" % (short, d / n, g / n))
