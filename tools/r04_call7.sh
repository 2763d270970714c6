cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04g_tl -- python3 $R/tools/per_frame.py --frames 400 > $R/gpurun_out/r04g_tl.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04g_tl_nosync -- python3 $R/tools/per_frame.py --frames 400 --no-sync > $R/gpurun_out/r04g_tl2.log 2>&1
cd $R
python3 tools/timeline.py gpurun_out/r04g_tl 2>&1 | tee gpurun_out/r04g_timeline.txt
python3 tools/timeline.py gpurun_out/r04g_tl_nosync 2>&1 | tee -a gpurun_out/r04g_timeline.txt
find gpurun_out/r04g_tl* -name "*.db" -delete
timeout -k 10 200 python tools/sweep.py --scene c3 --frames 256 --spp 256 defaults fold=0 fold=0,queue=0 2>&1 | tee gpurun_out/r04g_sweep.log
