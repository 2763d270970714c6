set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
timeout -k 10 600 python -m pytest tests/test_local_majorant_gpu.py -m gpu -q -x -k "pool_equals or scheduling or pure_function" > gpurun_out/r04F_tests.log 2>&1 || { tail -30 gpurun_out/r04F_tests.log; exit 1; }
tail -3 gpurun_out/r04F_tests.log
for v in "" _lmdnocold "" _lmdnocold; do echo "== lib$v" | tee -a gpurun_out/r04F_lm.log; for d in 2 4; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04F_lm.log; done; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3n --depth 2 --frames 128 --spp 128 lm=1 2>&1 | tee -a gpurun_out/r04F_lm.log; done
