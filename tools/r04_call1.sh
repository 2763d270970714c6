cd $GRAFT_REPO_ROOT
tools/calibrate_gather.sh r04 > gpurun_out/r04a_calib.log 2>&1
for lay in 2 3 4; do timeout -k 10 300 python tools/sweep.py --scene c3n --frames 256 --spp 256 --layout $lay lm=1 defaults 2>&1 | tee -a gpurun_out/r04a_layouts.log; done
for lay in 2 4; do timeout -k 10 300 python tools/sweep.py --scene c5 --frames 256 --spp 256 --layout $lay lm=1 defaults 2>&1 | tee -a gpurun_out/r04a_layouts.log; done
tail -40 gpurun_out/r04a_calib.log
