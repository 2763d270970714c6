set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_local_majorant_gpu.py -m gpu -q -x -k "pool_equals or pure_function or scheduling or means_agree" > gpurun_out/r04n_tests.log 2>&1 || { tail -30 gpurun_out/r04n_tests.log; exit 1; }
tail -3 gpurun_out/r04n_tests.log
for sc in c3 c3n c5 c2; do timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04n_lm.log; done
