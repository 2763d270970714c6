#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time from the kernel trace and
per-kernel, per-dispatch averages of every collected PMC counter."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    for k in ("k_split_front", "k_split_machine", "k_trace_env", "k_trace_tile", "k_trace_lm", "k_pathtrace_uloop", "k_pathtrace_pixel", "k_resolve", "k_tonemap", "k_raycast", "k_repack",
              "k_minmax", "k_empty_mask"):
        if k in name:
            return k + ("<brick>" if ("ILi2E" in name or "<2," in name) else "<linear>" if ("ILi1E" in name or "<1," in name) else "")
    return name[-60:]


print(f"# {out}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel stats (rocprofv3 --kernel-trace --stats)")
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print(f"{short(row['Name']):40s} calls={row['Calls']:>5s} total_ns={row['TotalDurationNs']:>12s} "
                  f"avg_ns={float(row['AverageNs']):>12.0f} min_ns={row['MinNs']:>10s} max_ns={row['MaxNs']:>10s} pct={row['Percentage']}")

acc = defaultdict(lambda: defaultdict(float))
ndisp = defaultdict(lambda: defaultdict(set))
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            c = row["Counter_Name"]
            acc[k][c] += float(row["Counter_Value"])
            ndisp[k][c].add(row["Dispatch_Id"])
print("## PMC counters, average per dispatch")
for k in sorted(acc):
    if not (k.startswith("k_pathtrace") or k.startswith("k_raycast") or k.startswith("k_trace") or k.startswith("k_split")):
        continue
    print(f"[{k}]")
    for c in sorted(acc[k]):
        n = max(1, len(ndisp[k][c]))
        print(f"  {c:40s} {acc[k][c] / n:18.1f}   (dispatches {n})")
