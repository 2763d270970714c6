set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_all.sh r04 2>&1 | tee gpurun_out/r04_profile_all.log | cut -c1-200
SPECS="c3_d4_split:--scene c3 --trace-depth 4 --set split=1" bash tools/profile_all.sh r04 2>&1 | tee gpurun_out/r04_profile_split.log | cut -c1-200
