set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_more_gpu.py -m gpu -q -x -k "parity or queue_machine or trips or pooled" > gpurun_out/r04A_tests.log 2>&1 || { tail -30 gpurun_out/r04A_tests.log; exit 1; }
tail -3 gpurun_out/r04A_tests.log
for v in "" _noldsl "" _noldsl; do echo "== lib$v" | tee -a gpurun_out/r04A_ldsl.log; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04A_ldsl.log; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth 2 --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04A_ldsl.log; for sc in c3n c5; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04A_ldsl.log; done; done
