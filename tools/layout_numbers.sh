#!/bin/bash
# PAIR vs CELL layout on the bench workloads: tools/layout_numbers.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/${1:-layout}_layouts.log
: > $OUT
for lay in 3 4; do
  echo "## layout $lay" | tee -a $OUT
  timeout -k 10 300 python tools/sweep.py --scene c3 --layout $lay --frames 256 --spp 256 defaults empty_skip=0 lm=1 queue=0 2>&1 | tee -a $OUT
  timeout -k 10 300 python tools/sweep.py --scene c3n --layout $lay --frames 256 --spp 256 defaults lm=1 2>&1 | tee -a $OUT
  timeout -k 10 300 python tools/sweep.py --scene c3 --depth 4 --layout $lay --frames 128 --spp 128 defaults 2>&1 | tee -a $OUT
done
