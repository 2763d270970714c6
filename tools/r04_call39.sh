set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/sweep.py --scene c3 --frames 256 --spp 256 defaults park_cheap=8 park_cheap=12 park_cheap=24 park_cheap=32 defaults 2>&1 | tee gpurun_out/r04I.log
timeout -k 10 300 python tools/sweep.py --scene c3 --depth 2 --frames 256 --spp 256 defaults park_end=16 park_end=24 park_end=40 park_end=48 defaults 2>&1 | tee -a gpurun_out/r04I.log
timeout -k 10 300 python tools/sweep.py --scene c3 --depth 4 --frames 256 --spp 256 defaults park_end=16 park_end=24 park_end=40 park_end=48 2>&1 | tee -a gpurun_out/r04I.log
timeout -k 10 300 python tools/sweep.py --scene c5 --frames 256 --spp 256 defaults park_cheap=8 park_cheap=24 2>&1 | tee -a gpurun_out/r04I.log
