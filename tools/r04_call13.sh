set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_env_nee_gpu.py -m gpu -q -x > gpurun_out/r04l_tests.log 2>&1 || { tail -30 gpurun_out/r04l_tests.log; exit 1; }
tail -3 gpurun_out/r04l_tests.log
timeout -k 10 300 python tools/sweep.py --scene c3 --depth 3 --frames 128 --spp 128 defaults env_nee=1 2>&1 | tee gpurun_out/r04l_env.log
