set -e
cd $GRAFT_REPO_ROOT
SPECS="c3_d4_split:--scene c3 --trace-depth 4 --set split=1" bash tools/profile_all.sh r04 2>&1 | tee gpurun_out/r04_profile_split.log | cut -c1-300
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
cat gpurun_out/r04_bench_default.json | cut -c1-600
bash tools/final_numbers.sh r04_final > /dev/null 2>&1
tail -5 gpurun_out/r04_final_numbers.log
