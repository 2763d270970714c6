#!/bin/bash
# One GPU-box call: the GPU test suite, then (unless a step was killed) the default bench.  Logs under gpurun_out/.
# Usage: tools/gpu_ci.sh <tag> [pytest-args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q --durations=15 "$@" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -25 gpurun_out/${TAG}_tests.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then echo "tests were killed: no further GPU step"; exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
brc=$?
echo "bench rc=$brc"; tail -3 gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.json
exit $(( rc != 0 ? rc : brc ))
