set -e
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
tail -1 gpurun_out/r04_bench_default.json | cut -c1-900
