cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04e_tl -- python3 $R/tools/per_frame.py --frames 400 > $R/gpurun_out/r04e_tl.log 2>&1
cd $R
python3 tools/timeline.py gpurun_out/r04e_tl 2>&1 | tee gpurun_out/r04e_timeline.txt
find gpurun_out/r04e_tl -name "*.db" -delete; find gpurun_out/r04e_tl -name "*kernel_trace.csv" -size +20M -delete
