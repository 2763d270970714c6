set -e
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/r04E_sweeps.log
timeout -k 10 300 python tools/sweep.py --scene c3n --depth 1 --frames 256 --spp 256 park_cheap=8 park_cheap=16 park_cheap=24 park_cheap=32 park_cheap=48 2>&1 | tee -a $LOG
timeout -k 10 300 python tools/sweep.py --scene c3n --depth 2 --frames 128 --spp 128 park_end=16 park_end=32 park_end=48 2>&1 | tee -a $LOG
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee -a $LOG
