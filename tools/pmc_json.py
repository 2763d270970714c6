#!/usr/bin/env python3
"""Turn a tools/profile.sh summary into the per-launch PMC record bench.py reads (profiles/r04_pmc_<tag>.json).
Usage: tools/pmc_json.py <summary.txt> <kernel-prefix>[+<kernel-prefix>...] <tag> <out.json> [note]
Counters are per-dispatch averages of separate rocprofv3 --pmc passes; several prefixes joined by '+' are kernels that are each dispatched
once per launch (the split kernels of deeper paths: k_split_front+k_split_machine): their counters and durations are summed.
FETCH_SIZE is doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section).  The correction was calibrated for THIS code's access
pattern (16-byte gathers, one per lane: profiles/r04_fetch_size_calibration.txt): one TCC_EA0_RDREQ per missing gather, tallied at 64 B,
while a miss moves the whole 128-byte line (whole-line and half-line random reads run at the same request rate) -- so the factor 2
holds for gathers as for streams; `fetch_correction` records which factor was used and why."""
import json
import re
import sys
from pathlib import Path

summary, kernel, tag, out = sys.argv[1:5]
note = sys.argv[5] if len(sys.argv) > 5 else ""
ROOT = Path(__file__).resolve().parents[1]
vals, cur, stats = {}, None, {}
kernels = kernel.split("+")
for line in open(summary):
    m = re.match(r"\[(.+)\]", line.strip())
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"\s+(\w+)\s+([0-9.eE+-]+)\s+\(dispatches (\d+)\)", line)
    if m and cur and any(cur.startswith(k) for k in kernels):
        vals[m.group(1)] = vals.get(m.group(1), 0.0) + float(m.group(2))
        continue
    m = re.match(r"(\S+)\s+calls=\s*(\d+)\s+total_ns=\s*(\d+)\s+avg_ns=\s*(\d+)", line)
    if m and any(m.group(1).lstrip(":").startswith(k) for k in kernels):
        stats = {"calls": int(m.group(2)), "avg_ns": stats.get("avg_ns", 0) + int(m.group(4))}
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
                "fetch_correction": {"factor": 2.0, "basis": "profiles/r04_fetch_size_calibration.txt: random 16-byte gathers = 1 TCC_EA0_RDREQ each, FETCH_SIZE = 64 B x RDREQ, "
                                                             "whole 128-byte lines and 64-byte half lines gathered at the same 54 G requests/s: a miss moves a line"},
                "traffic_bytes_per_launch": int(round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024))})
if "SQ_INSTS_VALU" in vals:
    rec["valu_insts_per_launch"] = int(vals["SQ_INSTS_VALU"])          # wave64 vector instructions
    rec["salu_insts_per_launch"] = int(vals.get("SQ_INSTS_SALU", 0))
    if "SQ_THREAD_CYCLES_VALU" in vals:
        rec["valu_lane_utilisation"] = round(vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_INSTS_VALU"] / 64.0, 4)
if "TCC_EA0_RDREQ_sum" in vals:
    rec["read_requests_per_launch"] = int(vals["TCC_EA0_RDREQ_sum"])     # memory-side read requests (lines); the gather ceiling is ~54 G of them per second
if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
    rec["l2_hit_rate"] = round(vals["TCC_HIT_sum"] / max(1.0, vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4)
if "TCP_TOTAL_CACHE_ACCESSES_sum" in vals and "TCP_TCC_READ_REQ_sum" in vals:
    rec["l1_hit_rate_reads"] = round(1.0 - vals["TCP_TCC_READ_REQ_sum"] / max(1.0, vals["TCP_TOTAL_CACHE_ACCESSES_sum"]), 4)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
