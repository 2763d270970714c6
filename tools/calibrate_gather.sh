#!/bin/bash
# Calibration of the memory-side counters for 16-byte random gathers (tools/ubench/gather16.hip): timings, then separate
# rocprofv3 --pmc passes (never combined with tracing).  usage: tools/calibrate_gather.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r04}_gather_calibration
mkdir -p $OUT
BIN=$ROOT/tools/ubench/gather16
cd /tmp && export TMPDIR=/tmp
{
echo "## timings (HIP events, best of 3): table MiB / ilp / gathers per lane / blocks per CU / pattern"
for mib in 200 2200; do
  for ilp in 1 2 4 8; do timeout -k 10 120 $BIN $mib $ilp 512 1 0; done
  timeout -k 10 120 $BIN $mib 4 512 2 0
  timeout -k 10 120 $BIN $mib 1 512 1 1
  timeout -k 10 120 $BIN $mib 4 512 1 1
  timeout -k 10 120 $BIN $mib 1 256 1 2
done
timeout -k 10 120 $BIN 17000 4 512 1 0
timeout -k 10 120 $BIN 17000 1 512 1 0
} 2>&1 | tee $OUT/timings.txt
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  for cfg in "2200 4 256 1 0" "200 4 256 1 0" "2200 4 256 1 1" "17000 4 256 1 0"; do
    tag=$(echo $cfg | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $group --output-format csv -d $OUT/pmc_${i}_$tag -- $BIN $cfg > $OUT/pmc_${i}_$tag.log 2>&1 || { echo "pmc $i $tag failed"; tail -3 $OUT/pmc_${i}_$tag.log; }
  done
done <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum
GROUPS
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    cfg = f.split("/pmc_")[1].split("/")[0].split("_", 1)[1]
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "k_gather16" in row["Kernel_Name"]:
                acc[cfg][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
print("## PMC per timed dispatch of k_gather16 (the last 3 dispatches of a run are the timed ones: 256 gathers per lane, 256 CUs x 1024 lanes = 67 108 864 requests of 16 B)")
for cfg in sorted(acc):
    print(f"[{cfg}]  (table MiB, ilp, gathers per lane, blocks per CU, pattern)")
    for c in sorted(acc[cfg]):
        byd = defaultdict(float)
        for d, v in acc[cfg][c]:
            byd[d] += v
        ds = sorted(byd)[-3:]
        print(f"  {c:36s} " + "  ".join(f"{byd[d]:16.1f}" for d in ds))
PY
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*agent_info.csv" -delete 2>/dev/null
