#!/usr/bin/env python3
"""Summarises two rocprofv3 counter_collection CSVs (FETCH_SIZE pass, WRITE_SIZE
pass) into HBM bytes per kernel launch.  Units: the counters are in KiB
(x1024).  Calibration: bench.py --calib-copy runs one torch copy of a known
size (1 GiB read + 1 GiB written); its counters give the correction factors for
this access width (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 on gfx950)."""
import csv
import json
import sys
from collections import defaultdict

CALIB_BYTES = 1 << 30


def load(path):
    per = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return per


def short(name):
    for key, tag in (("k_fused", "filters"), ("k_filters", "filters"), ("k_xyb_only", "filters"), ("k_epf0", "epf0"), ("k_transform_mfma32", "blocks_mfma32"), ("k_transform_8", "blocks_8x8"),
                     ("k_transform_r16", "blocks_r16"), ("k_transform_r32", "blocks_r32"), ("k_transform_r", "blocks_r"),
                     ("k_transform_a", "blocks_a"), ("k_large", "blocks_large"), ("k_prepare", "prepare")):
        if key in name:
            return tag
    return None


fetch, write = load(sys.argv[1]), load(sys.argv[2])


def calib(per):
    # the calibration copy is the single largest non-jxlhip dispatch
    best = 0.0
    for k, v in per.items():
        if "jxlhip" in k:
            continue
        best = max(best, max(v))
    return best * 1024.0


cf, cw = calib(fetch), calib(write)
ff = CALIB_BYTES / cf if cf else 2.0
fw = CALIB_BYTES / cw if cw else 1.0
out = {"_calibration": {"copy_bytes_each_way": CALIB_BYTES, "raw_fetch_bytes": cf, "raw_write_bytes": cw,
                        "fetch_factor": round(ff, 4), "write_factor": round(fw, 4)}}
agg = defaultdict(lambda: [0.0, 0.0, 0])
for k, v in fetch.items():
    s = short(k)
    if s:
        agg[s][0] += sum(v) * 1024.0 * ff
        agg[s][2] = max(agg[s][2], len(v))
for k, v in write.items():
    s = short(k)
    if s:
        agg[s][1] += sum(v) * 1024.0 * fw
launch = {}
for s, (fb, wb, n) in agg.items():
    launch[s] = {"read_bytes": round(fb / n), "write_bytes": round(wb / n), "launches": n}
out["per_launch_detail"] = launch
# totals in bench.py's kernel slots
slots = defaultdict(float)
for s, d in launch.items():
    slots[s.split(":")[0]] += d["read_bytes"] + d["write_bytes"]
out.update({k: round(v) for k, v in slots.items()})
print(json.dumps(out, indent=1))
