set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
for v in "" _nofb "" _nofb; do echo "== lib$v" | tee -a gpurun_out/r04v_lm.log; for sc in c3 c3n c5; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04v_lm.log; done; done
