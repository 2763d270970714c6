set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
for v in "" _lm_ilp _lm_mc _lm_minreg _lm_nounroll _lm_o2 ""; do echo "== lib$v" | tee -a gpurun_out/r04x_lm.log; for sc in c5 c3 c3n; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04x_lm.log; done; done
