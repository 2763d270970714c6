cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
timeout -k 10 300 python tools/per_frame.py 2>&1 | tee gpurun_out/r04d_per_frame.log
timeout -k 10 300 python tools/per_frame.py --no-sync 2>&1 | tee -a gpurun_out/r04d_per_frame.log
timeout -k 10 300 python tools/per_frame.py queue=0 2>&1 | tee -a gpurun_out/r04d_per_frame.log
timeout -k 10 300 python tools/per_frame.py --depth 2 2>&1 | tee -a gpurun_out/r04d_per_frame.log
timeout -k 10 300 python tools/per_frame.py --scene c3n 2>&1 | tee -a gpurun_out/r04d_per_frame.log
for sc in c3 c3n c5; do SVR_HIP_LIB=$L/libsvr_hip_prof.so timeout -k 10 300 python tools/lm_phase_prof.py --scene $sc 2>&1 | tee -a gpurun_out/r04d_lm_phase.log; done
for v in "" _b8 _b4; do for sc in c3 c3n c5; do echo "== lib$v"; SVR_HIP_LIB=$L/libsvr_hip$v.so timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 2>&1 | tee -a gpurun_out/r04d_lm_batch.log; done; done
