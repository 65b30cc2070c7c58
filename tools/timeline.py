#!/usr/bin/env python3
"""GPU box: where a decode step's time goes between its launches.  usage: tools/timeline.py <kernel_trace.csv> [first kernel substring]
Reads rocprofv3 --kernel-trace output, cuts the stream of launches into steps at every `k_prepare`, and prints the
average duration of each launch and of the gap in front of it (end of the previous launch -> its start)."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
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
steps = [st for st in steps if len(st) == max(len(x) for x in steps)][5:-1]  # steady state, complete steps
acc = defaultdict(lambda: [0.0, 0.0, 0])
period = []
for i, st in enumerate(steps):
    for j, (s, e, n) in enumerate(st):
        short = n.split("(")[0].replace("void jxlhip::", "").replace("(anonymous namespace)::", "")[:60]
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
