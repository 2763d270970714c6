set -e
cd $GRAFT_REPO_ROOT
T=$(( 1 + (16<<8) + (16<<16) ))
for sc in c3 c5; do timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 lm=1,lm_tune=$(( T + (6<<24) )) lm=1,lm_tune=$(( T + (8<<24) )) lm=1,lm_tune=$(( T + (12<<24) )) lm=1,lm_tune=$(( T + (14<<24) )) 2>&1 | tee -a gpurun_out/r04p_lm.log; done
T=$(( 3 + (24<<8) + (24<<16) ))
timeout -k 10 300 python tools/sweep.py --scene c3n --frames 256 --spp 256 lm=1 lm=1,lm_tune=$(( T + (18<<24) )) lm=1,lm_tune=$(( T + (23<<24) )) 2>&1 | tee -a gpurun_out/r04p_lm.log
timeout -k 10 600 python -m pytest tests/test_local_majorant_gpu.py -m gpu -q -x -k "pool_equals or scheduling" > gpurun_out/r04p_tests.log 2>&1 || { tail -30 gpurun_out/r04p_tests.log; exit 1; }
tail -3 gpurun_out/r04p_tests.log
