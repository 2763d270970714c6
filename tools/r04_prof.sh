cd $GRAFT_REPO_ROOT
bash tools/profile_all.sh r04 2>&1 | tee gpurun_out/r04_profile_all.log
