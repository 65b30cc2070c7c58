#!/usr/bin/env python3
"""Prints, from a rocprofv3 --kernel-trace CSV, the launches of the LAST frames of a run as a timeline: start and end
of every dispatch relative to the frame's first k_prepare, the queue it ran on, and which dispatches overlap.
usage: trace_timeline.py <kernel_trace.csv> [frames=2]"""
import csv
import re
import sys


def main():
    path = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name") or r.get("kernel_name")
            m = re.search(r"\b(k_[a-z0-9_]+)", name)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else name[:40], r.get("Queue_Id", "?")))
    rows.sort()
    rows = [r for r in rows if r[2].startswith("k_")]  # the library's launches only
    starts = [i for i, r in enumerate(rows) if "k_prepare" in r[2]]
    # frame = from a k_prepare that follows a fused kernel (or the first) up to the next such
    heads = [i for k, i in enumerate(starts) if k == 0 or any("k_fused" in rows[j][2] for j in range(starts[k - 1], i))]
    for h, nxt in list(zip(heads, heads[1:] + [len(rows)]))[-frames:]:
        t0 = rows[h][0]
        print(f"frame at +{(t0 - rows[0][0]) / 1e3:.1f} us, {nxt - h} dispatches, span {(max(r[1] for r in rows[h:nxt]) - t0) / 1e3:.1f} us")
        for s, e, n, q in rows[h:nxt]:
            print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f}  ({(e - s) / 1e3:7.1f} us)  q{q}  {n}")


if __name__ == "__main__":
    main()
