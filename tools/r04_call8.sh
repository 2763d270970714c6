cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04h_tl -- python3 $R/tools/ahead_alone.py c3 > $R/gpurun_out/r04h_tl.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04h_tl_q0 -- python3 $R/tools/ahead_alone.py c3 queue=0 > $R/gpurun_out/r04h_tl2.log 2>&1
cd $R
python3 tools/timeline.py gpurun_out/r04h_tl 2>&1 | grep -A8 "k_trace_tile\|last trace" | tee gpurun_out/r04h_timeline.txt
python3 tools/timeline.py gpurun_out/r04h_tl_q0 2>&1 | grep -A8 "k_trace_tile\|last trace" | tee -a gpurun_out/r04h_timeline.txt
find gpurun_out/r04h_tl* -name "*.db" -delete
