#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py [svr_trace_tile.hip] [name-filter]"""
import re
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import _build  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "svr_trace_tile.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [_build._hipcc(), *_build.HIPCC_FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(_build.CSRC / src), "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
name, row = None, {}
for line in out.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        name, row = m.group(1), {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and name:
        row[m.group(1)] = int(m.group(2))
        if m.group(1).startswith("LDS"):
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            if flt in dem:
                print(f"{dem[:70]:70s} vgpr {row.get('VGPRs')} agpr {row.get('AGPRs')} spill {row.get('VGPRs Spill')} sgpr-spill {row.get('SGPRs Spill')} scratch {row.get('ScratchSize [bytes/lane]')} occ {row.get('Occupancy [waves/SIMD]')} lds {row.get('LDS Size [bytes/block]')}")
