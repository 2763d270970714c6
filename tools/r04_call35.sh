set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_more_gpu.py tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "not frame_ahead_steady" > gpurun_out/r04E_tests.log 2>&1 || { tail -30 gpurun_out/r04E_tests.log; exit 1; }
tail -3 gpurun_out/r04E_tests.log
for d in 2 3 4 6; do timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 defaults split=0 2>&1 | tee -a gpurun_out/r04E_split.log; done
timeout -k 10 300 python tools/sweep.py --scene c5 --depth 2 --frames 128 --spp 128 defaults split=0 2>&1 | tee -a gpurun_out/r04E_split.log
timeout -k 10 300 python tools/sweep.py --scene c3b --depth 4 --frames 128 --spp 128 defaults split=0 2>&1 | tee -a gpurun_out/r04E_split.log
