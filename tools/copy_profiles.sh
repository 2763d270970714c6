#!/bin/bash
# Copy what tools/profile_all.sh left under gpurun_out/prof_<tag>_* into profiles/: the PMC record bench.py reads, the summary and rocprof's kernel-stats table.
# usage: tools/copy_profiles.sh <round-tag>
set -eu
R=$1
cd "$(dirname "$0")/.."
for d in gpurun_out/prof_${R}_*/; do
  t=$(basename $d); t=${t#prof_${R}_}
  [ -f $d/pmc.json ] || continue
  cp $d/pmc.json profiles/${R}_pmc_$t.json
  cp $d/summary.txt profiles/${R}_${t}_prof_summary.txt
  ks=$(find $d/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$ks" ] && cp $ks profiles/${R}_${t}_kernel_stats.csv
done
ls profiles | grep -c "^${R}_pmc_"
