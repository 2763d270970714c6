cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_more_gpu.py -m gpu -q -x -k "frame_ahead or queue_machine or pooled or nan_guard" > gpurun_out/r04c_tests.log 2>&1; tail -5 gpurun_out/r04c_tests.log
for m in 200 2200; do for p in 3 4; do tools/ubench/gather16 $m 4 512 1 $p; done; done 2>&1 | tee gpurun_out/r04c_gather_lines.log
timeout -k 10 300 python tools/per_frame.py 2>&1 | tee gpurun_out/r04c_per_frame.log
timeout -k 10 300 python tools/per_frame.py --no-sync 2>&1 | tee -a gpurun_out/r04c_per_frame.log
timeout -k 10 300 python tools/per_frame.py queue=0 2>&1 | tee -a gpurun_out/r04c_per_frame.log
timeout -k 10 300 python tools/per_frame.py --depth 2 2>&1 | tee -a gpurun_out/r04c_per_frame.log
timeout -k 10 300 python tools/per_frame.py --scene c3n 2>&1 | tee -a gpurun_out/r04c_per_frame.log
timeout -k 10 300 python tools/sweep.py --scene c3 --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04c_per_frame.log
