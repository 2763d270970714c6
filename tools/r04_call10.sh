set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_more_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "queue_machine or parity or trips or non_cubic" > gpurun_out/r04j_tests.log 2>&1 || { tail -20 gpurun_out/r04j_tests.log; exit 1; }
tail -3 gpurun_out/r04j_tests.log
timeout -k 10 300 python tools/sweep.py --scene c3 --depth 2 --frames 64 --spp 64 defaults 2>&1 | tee gpurun_out/r04j_split.log
for d in 2 4 6; do timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 defaults split=0 2>&1 | tee -a gpurun_out/r04j_split.log; done
timeout -k 10 300 python tools/sweep.py --scene c5 --depth 2 --frames 128 --spp 128 defaults split=0 2>&1 | tee -a gpurun_out/r04j_split.log
timeout -k 10 300 python tools/sweep.py --scene c3b --depth 4 --frames 128 --spp 128 defaults split=0 2>&1 | tee -a gpurun_out/r04j_split.log
