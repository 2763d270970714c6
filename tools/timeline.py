#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run (csv): per kernel name count / mean duration, and for the steady state of
tools/per_frame.py how the per-call resolves sit relative to the trace launches.  usage: tools/timeline.py <dir>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Stream_Id", ""), r.get("Queue_Id", "")))
rows.sort()
t0 = rows[0][0]
from collections import defaultdict
d = defaultdict(list)
for s, e, n, st, q in rows:
    d[n].append(e - s)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:50s} n={len(v):6d} mean={sum(v)/len(v)/1e3:9.1f} us  total={sum(v)/1e6:9.2f} ms")
# last 3 trace launches and the resolves around them
tr = [(s, e) for s, e, n, st, q in rows if "k_trace_tile" in n]
print("last trace launches (start, end, duration ms, gap to previous end ms):")
for i in range(max(1, len(tr) - 6), len(tr)):
    print(f"  {(tr[i][0]-t0)/1e6:10.3f} {(tr[i][1]-t0)/1e6:10.3f} {(tr[i][1]-tr[i][0])/1e6:8.3f} {(tr[i][0]-tr[i-1][1])/1e6:8.3f}")
if len(tr) >= 3:
    a, b = tr[-3][0], tr[-2][1]
    rs = [(s, e) for s, e, n, st, q in rows if ("k_resolve" in n or "k_mean_flat" in n) and a <= s <= b]
    tm = [(s, e) for s, e, n, st, q in rows if "k_tonemap" in n and a <= s <= b]
    if tm:
        print(f"  tone maps in that window: {len(tm)}, mean duration {sum(e-s for s,e in tm)/len(tm)/1e3:.1f} us")
    inside = [(s, e) for s, e in rs if any(ts <= s and e <= te for ts, te in tr)]
    print(f"resolves started between the last-but-two trace start and the last-but-one trace end: {len(rs)}, of which inside a trace launch: {len(inside)}")
    if rs:
        print(f"  mean resolve duration {sum(e-s for s,e in rs)/len(rs)/1e3:.1f} us; inside-trace mean {sum(e-s for s,e in inside)/max(1,len(inside))/1e3:.1f} us")
        gaps = [rs[i+1][0]-rs[i][1] for i in range(len(rs)-1)]
        print(f"  mean gap between consecutive resolves {sum(gaps)/max(1,len(gaps))/1e3:.1f} us, max {max(gaps)/1e3:.1f} us")
