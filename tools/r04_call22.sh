set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r04r_tests.log 2>&1 || { tail -40 gpurun_out/r04r_tests.log; exit 1; }
tail -18 gpurun_out/r04r_tests.log
for lay in 2 3 4; do timeout -k 10 300 python tools/sweep.py --scene c3n --frames 256 --spp 256 --layout $lay defaults 2>&1 | tee -a gpurun_out/r04r_c3n_layouts.log; done
