cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_more_gpu.py -m gpu -q -x -k "frame_ahead or nan_guard or queue_machine" 2>&1 | tail -3
timeout -k 10 300 python tools/per_frame.py 2>&1 | tee gpurun_out/r04i_per_frame.log
timeout -k 10 300 python tools/per_frame.py --depth 2 2>&1 | tee -a gpurun_out/r04i_per_frame.log
timeout -k 10 300 python tools/per_frame.py --depth 4 2>&1 | tee -a gpurun_out/r04i_per_frame.log
timeout -k 10 300 python tools/per_frame.py --scene c3n 2>&1 | tee -a gpurun_out/r04i_per_frame.log
timeout -k 10 300 python tools/per_frame.py --scene c5 2>&1 | tee -a gpurun_out/r04i_per_frame.log
timeout -k 10 200 python tools/sweep.py --scene c3 --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04i_per_frame.log
