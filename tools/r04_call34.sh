set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
for v in "" _mldsl _mboth "" _mldsl _mboth; do echo "== lib$v" | tee -a gpurun_out/r04D_split.log; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth 2 --frames 256 --spp 256 split=2 split=0 2>&1 | tee -a gpurun_out/r04D_split.log; for d in 3 4; do SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 defaults 2>&1 | tee -a gpurun_out/r04D_split.log; done; done
