set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
timeout -k 10 600 python -m pytest tests/test_more_gpu.py -m gpu -q -x -k "split" > gpurun_out/r04C_tests.log 2>&1 || { tail -30 gpurun_out/r04C_tests.log; exit 1; }
tail -3 gpurun_out/r04C_tests.log
for v in "" _spnocold "" _spnocold; do echo "== lib$v" | tee -a gpurun_out/r04C_split.log; for d in 3 4 6; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04C_split.log; done; done
