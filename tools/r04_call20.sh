set -e
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
T=$(( 1 + (16<<8) + (16<<16) ))
for sc in c3 c5; do timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 lm=1,lm_tune=$(( T + (8<<24) )) lm=1,lm_tune=$(( T + (12<<24) )) lm=1,lm_tune=$(( T + (16<<24) )) 2>&1 | tee -a gpurun_out/r04q_lm.log; SVR_HIP_LIB=$L/libsvr_hip_v10s.so timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04q_lm.log; done
T=$(( 3 + (24<<8) + (24<<16) ))
timeout -k 10 300 python tools/sweep.py --scene c3n --frames 256 --spp 256 lm=1 lm=1,lm_tune=$(( T + (16<<24) )) 2>&1 | tee -a gpurun_out/r04q_lm.log
SVR_HIP_LIB=$L/libsvr_hip_v21s.so timeout -k 10 300 python tools/sweep.py --scene c3n --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04q_lm.log
