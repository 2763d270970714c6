#!/bin/bash
# The three measured workloads of bench.py under rocprofv3 (kernel trace + separate PMC passes each), and the per-launch
# PMC records bench.py reads.  Usage: tools/profile_all.sh <round-tag>   (results under gpurun_out/prof_<tag>_*)
set -u
R=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export PMC_ONLY="${PMC_ONLY:-1 2 3 4 5 6}"
SPECS=${SPECS:-"c3_d1:--scene c3|c3_d1_noskip:--scene c3 --empty-skip 0|c3n_d1:--scene c3n|c3_d2:--scene c3 --trace-depth 2|c3_d4:--scene c3 --trace-depth 4|c3n_d4:--scene c3n --trace-depth 4|c5_d1:--scene c5|c3_d1_lm:--scene c3 --set local_majorant=1|c3n_d1_lm:--scene c3n --set local_majorant=1|c5_d1_lm:--scene c5 --set local_majorant=1|c3_d2_lm:--scene c3 --trace-depth 2 --set local_majorant=1|c3_d4_lm:--scene c3 --trace-depth 4 --set local_majorant=1"}
IFS='|' read -ra SPEC_LIST <<< "$SPECS"
for spec in "${SPEC_LIST[@]}"; do
  tag=${spec%%:*}; args=${spec#*:}
  # (the opt-in two-kernel form of deeper paths, --set split=1, would be kern=k_split_front+k_split_machine)
  kern=k_trace_tile; case $tag in *_lm) kern=k_trace_lm;; *_split) kern=k_split_front+k_split_machine;; esac
  echo "=== $tag ($args)"
  PROF_STEPS=${PROF_STEPS:-1} bash tools/profile.sh ${R}_$tag $args --spp-per-step ${PROF_SPP:-64} > gpurun_out/prof_${R}_$tag.log 2>&1 || { echo "profile $tag failed"; tail -5 gpurun_out/prof_${R}_$tag.log; exit 1; }
  python3 tools/pmc_json.py gpurun_out/prof_${R}_$tag/summary.txt "$kern" $tag gpurun_out/prof_${R}_$tag/pmc.json "bench.py $args --spp-per-step ${PROF_SPP:-64} (64 frames per launch)" | cut -c1-400
done
