set -e
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/r04F_lm_bytes.log
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
for v in "" _cls "" _cls; do
  echo "== lib$v" | tee -a $LOG
  for sc in c5 c3 c3n; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene $sc --depth 1 --frames 256 --spp 256 --count lm=1 2>&1 | tee -a $LOG; done
  SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth 2 --frames 256 --spp 256 lm=1 2>&1 | tee -a $LOG
  SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3n --depth 2 --frames 128 --spp 128 lm=1 2>&1 | tee -a $LOG
done
timeout -k 10 900 python -m pytest tests/test_local_majorant_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a $LOG
