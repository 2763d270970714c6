cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sunvolumerender_amd/lib
timeout -k 10 300 python -m pytest tests/test_more_gpu.py -m gpu -q -x -k "frame_ahead or nan_guard" 2>&1 | tail -3
timeout -k 10 300 python tools/per_frame.py 2>&1 | tee gpurun_out/r04f_per_frame.log
timeout -k 10 300 python tools/per_frame.py queue=0 2>&1 | tee -a gpurun_out/r04f_per_frame.log
timeout -k 10 300 python tools/per_frame.py --depth 2 2>&1 | tee -a gpurun_out/r04f_per_frame.log
for sc in c3 c3n c5; do SVR_HIP_LIB=$L/libsvr_hip_prof.so timeout -k 10 300 python tools/lm_phase_prof.py --scene $sc 2>&1 | tee -a gpurun_out/r04f_lm_phase.log; done
