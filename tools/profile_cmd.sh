#!/bin/bash
# PMC passes for an arbitrary python tool: tools/profile_cmd.sh <tag> <script> [args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/$*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; }
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  echo "== pmc $i: $group"
  timeout -k 10 240 rocprofv3 --pmc $group --output-format csv -d $OUT/pmc_$i -- $CMD > $OUT/pmc_$i.log 2>&1 || { echo "pmc $i failed"; tail -3 $OUT/pmc_$i.log; }
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
GROUPS
python3 $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*agent_info.csv" -delete 2>/dev/null
