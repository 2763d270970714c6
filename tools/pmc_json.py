#!/usr/bin/env python3
"""Turn a tools/profile.sh summary into the per-launch PMC record bench.py reads (profiles/r03_pmc_<tag>.json).
Usage: tools/pmc_json.py <summary.txt> <kernel-prefix> <tag> <out.json> [note]
Counters are per-dispatch averages of separate rocprofv3 --pmc passes; FETCH_SIZE is doubled per the gfx950
correction of MI355X_MICROARCH.md (HBM section)."""
import json
import re
import sys
from pathlib import Path

summary, kernel, tag, out = sys.argv[1:5]
note = sys.argv[5] if len(sys.argv) > 5 else ""
ROOT = Path(__file__).resolve().parents[1]
vals, cur, stats = {}, None, {}
for line in open(summary):
    m = re.match(r"\[(.+)\]", line.strip())
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"\s+(\w+)\s+([0-9.eE+-]+)\s+\(dispatches (\d+)\)", line)
    if m and cur and cur.startswith(kernel):
        vals[m.group(1)] = float(m.group(2))
        continue
    m = re.match(r"(\S+)\s+calls=\s*(\d+)\s+total_ns=\s*(\d+)\s+avg_ns=\s*(\d+)", line)
    if m and m.group(1).startswith(kernel):
        stats = {"calls": int(m.group(2)), "avg_ns": int(m.group(4))}
sys.path.insert(0, str(ROOT))
from sunvolumerender_amd._build import kernel_source_hash  # noqa: E402
rec = {"source": f"{summary} (rocprofv3 --kernel-trace --stats, then separate --pmc passes; per-dispatch averages)", "kernel": kernel, "tag": tag,
       "note": note, "kernel_source_hash": kernel_source_hash(),
       # what bench.py calls the default launch shape: the only one it uses the record for
       "launch_shape": {"frames_per_launch": 64, "layout": "auto", "queue": "auto", "blocks_per_cu": "default", "extra_options": []}}
if stats:
    rec["rocprof_kernel_avg_ms"] = stats["avg_ns"] / 1e6
    rec["rocprof_kernel_calls"] = stats["calls"]
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    rec.update({"FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
                "correction": "gfx950: FETCH_SIZE reports half of the read bytes -> doubled (MI355X_MICROARCH.md, HBM)",
                "traffic_bytes_per_launch": int(round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024))})
if "SQ_INSTS_VALU" in vals:
    rec["valu_insts_per_launch"] = int(vals["SQ_INSTS_VALU"])          # wave64 vector instructions
    rec["salu_insts_per_launch"] = int(vals.get("SQ_INSTS_SALU", 0))
    if "SQ_THREAD_CYCLES_VALU" in vals:
        rec["valu_lane_utilisation"] = round(vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_INSTS_VALU"] / 64.0, 4)
if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
    rec["l2_hit_rate"] = round(vals["TCC_HIT_sum"] / max(1.0, vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4)
if "TCP_TOTAL_CACHE_ACCESSES_sum" in vals and "TCP_TCC_READ_REQ_sum" in vals:
    rec["l1_hit_rate_reads"] = round(1.0 - vals["TCP_TCC_READ_REQ_sum"] / max(1.0, vals["TCP_TOTAL_CACHE_ACCESSES_sum"]), 4)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
