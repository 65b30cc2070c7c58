#!/usr/bin/env python3
"""Prints average kernel durations (us) from a rocprofv3 kernel_stats.csv; args: file [substr...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    n = r["Name"]
    if "jxlhip" not in n:
        continue
    s = n.split("(jxlhip::DevFrame")[0].replace("void jxlhip::", "").replace("(anonymous namespace)::", "")
    if pats and not any(p in s for p in pats):
        continue
    print(f"  {s:40s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1000:8.1f}")
