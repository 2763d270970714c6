set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_more_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "split or queue_machine or parity or trips" > gpurun_out/r04k_tests.log 2>&1 || { tail -20 gpurun_out/r04k_tests.log; exit 1; }
tail -3 gpurun_out/r04k_tests.log
for d in 2 3 4; do timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 defaults split=0 split=2 2>&1 | tee -a gpurun_out/r04k_split.log; done
