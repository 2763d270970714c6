#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE averages of a tools/profile.sh summary into the per-launch HBM traffic
record bench.py reads (profiles/r01_traffic_<kernel>_<scene>_s<spp>.json).
Usage: tools/traffic_json.py <summary.txt> <kernel-prefix> <scene> <spp_per_step> <out.json>"""
import json
import re
import sys

summary, kernel, scene, spp, out = sys.argv[1:6]
vals, cur = {}, None
for line in open(summary):
    m = re.match(r"\[(.+)\]", line.strip())
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"\s+(\w+)\s+([0-9.eE+-]+)\s+\(dispatches (\d+)\)", line)
    if m and cur and cur.startswith(kernel):
        vals[m.group(1)] = float(m.group(2))
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
rec = {
    "source": f"{summary} (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --spp-per-step {spp} --steps 4)",
    "kernel": kernel, "scene": scene, "spp_per_step": int(spp),
    "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE reports half of the read bytes -> doubled (MI355X_MICROARCH.md, HBM)",
    "traffic_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)),
}
if "SQ_INSTS_VALU" in vals:
    rec["valu_insts_per_launch"] = int(vals["SQ_INSTS_VALU"])          # wave64 vector instructions
    rec["salu_insts_per_launch"] = int(vals.get("SQ_INSTS_SALU", 0))
    rec["valu_lane_utilisation"] = round(vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_INSTS_VALU"] / 64.0, 4) if "SQ_THREAD_CYCLES_VALU" in vals else None
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
