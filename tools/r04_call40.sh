set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
for v in "" _nb6 _nb8 _nb12 _nb14 ""; do echo "== lib$v" | tee -a gpurun_out/r04J.log; for sc in c5 c3; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04J.log; done; done
