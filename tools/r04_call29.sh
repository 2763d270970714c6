set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
timeout -k 10 600 python -m pytest tests/test_more_gpu.py -m gpu -q -x -k "window_and_a_row_shard or split_kernels" > gpurun_out/r04y_tests.log 2>&1 || { tail -30 gpurun_out/r04y_tests.log; exit 1; }
tail -3 gpurun_out/r04y_tests.log
for v in "" _tm "" _tm; do echo "== lib$v" | tee -a gpurun_out/r04y_field.log; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --frames 256 --spp 256 defaults fast_math=1 lm=1 2>&1 | tee -a gpurun_out/r04y_field.log; for d in 2 4; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04y_field.log; done; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3n --depth 2 --frames 128 --spp 128 defaults 2>&1 | tee -a gpurun_out/r04y_field.log; done
