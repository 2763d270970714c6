cd $GRAFT_REPO_ROOT
bash tools/final_numbers.sh r04 > /dev/null 2>&1
tail -5 gpurun_out/r04_numbers.log
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/r04_bench_default.err; head -c 2500 gpurun_out/r04_bench_default.json
