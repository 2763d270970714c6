#!/bin/bash
# Profile bench.py on the GPU box: one --kernel-trace --stats run, then separate --pmc passes
# (counters are never combined with tracing domains).  Usage: tools/profile.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/{trace,pmc_*}/...csv and a summary in gpurun_out/prof_<tag>/summary.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --cpu-seconds 0 --no-count --no-extra --steps ${PROF_STEPS:-2} --warmup 1 $*"
echo "== trace"
# (the kernel-trace pass is cheap: TRACE_STEPS timed steps after 2 warm-up steps, so that the average duration is not that of a cold launch)
TRACE="python3 $ROOT/bench.py --cpu-seconds 0 --no-count --no-extra --steps ${TRACE_STEPS:-8} --warmup 2 $*"
rm -rf $OUT/trace
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; }
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi
  echo "== pmc $i: $group"
  timeout -k 10 240 rocprofv3 --pmc $group --output-format csv -d $OUT/pmc_$i -- $BENCH > $OUT/pmc_$i.log 2>&1 || { echo "pmc $i failed"; tail -3 $OUT/pmc_$i.log; }
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
WRITE_SIZE
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
GROUPS
python3 $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep the merge small: drop everything but stats/summary/counter csv
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*agent_info.csv" -delete 2>/dev/null
