#!/bin/bash
# pool tuning sweep: cells per turn | refill << 8 | ended << 16
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/${1:-tune}_lm_tune.log
: > $OUT
for sc in c3 c3n c5; do
  timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 lm=1 lm=1,lm_sub=0 lm=1,lm_tune=0x101001 lm=1,lm_tune=0x101002 lm=1,lm_tune=0x101004 lm=1,lm_tune=0x0c0c03 lm=1,lm_tune=0x181803 lm=1,lm_tune=0x200c03 2>&1 | tee -a $OUT
done
