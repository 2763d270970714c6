set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_more_gpu.py tests/test_local_majorant_gpu.py -m gpu -q -x -k "frame_ahead or row_shard or pure_function" > gpurun_out/r04z_tests.log 2>&1 || { tail -30 gpurun_out/r04z_tests.log; exit 1; }
tail -3 gpurun_out/r04z_tests.log
timeout -k 10 300 python tools/per_frame.py 2>&1 | tee gpurun_out/r04z_per_frame.log
timeout -k 10 300 python tools/per_frame.py --depth 2 2>&1 | tee -a gpurun_out/r04z_per_frame.log
timeout -k 10 300 python tools/per_frame.py --scene c3n 2>&1 | tee -a gpurun_out/r04z_per_frame.log
