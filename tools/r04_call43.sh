set -e
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/r04E_fast_bound3.log
timeout -k 10 900 python -m pytest tests/test_more_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "fast_bound or trips or pooled or parity" 2>&1 | tail -5 | tee -a $LOG
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c3n" 2>&1 | tail -5 | tee -a $LOG
timeout -k 10 300 python tools/sweep.py --scene c3n --depth 1 --frames 256 --spp 256 fast_bound=1 fast_bound=0 fast_bound=1 fast_bound=0 2>&1 | tee -a $LOG
timeout -k 10 300 python tools/sweep.py --scene c3n --depth 2 --frames 128 --spp 128 fast_bound=1 fast_bound=0 2>&1 | tee -a $LOG
timeout -k 10 300 python tools/sweep.py --scene c3n --depth 4 --frames 128 --spp 128 fast_bound=1 fast_bound=0 2>&1 | tee -a $LOG
timeout -k 10 300 python tools/sweep.py --scene c3 --depth 1 --frames 512 --spp 256 defaults 2>&1 | tee -a $LOG
