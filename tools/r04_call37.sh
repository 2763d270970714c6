set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
for v in "" _cdp "" _cdp; do echo "== lib$v" | tee -a gpurun_out/r04G.log; for d in 2 4; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3n --depth $d --frames 128 --spp 128 defaults 2>&1 | tee -a gpurun_out/r04G.log; done; done
