set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
timeout -k 10 600 python -m pytest tests/test_more_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "parity or queue_machine or trips or pooled" > gpurun_out/r04H_tests.log 2>&1 || { tail -30 gpurun_out/r04H_tests.log; exit 1; }
tail -3 gpurun_out/r04H_tests.log
for v in "" _cda "" _cda; do echo "== lib$v" | tee -a gpurun_out/r04H.log; for d in 2 4; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 split=0 2>&1 | tee -a gpurun_out/r04H.log; done; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/per_frame.py --depth 2 2>&1 | tee -a gpurun_out/r04H.log; done
