#!/bin/bash
# Local-majorant mode vs the default (bit-exact) mode on the bench scenes: tools/lm_numbers.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/${1:-lm}_numbers.log
: > $OUT
for sc in c3 c3n c5 c2; do timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 --count defaults lm=1 2>&1 | tee -a $OUT; done
for d in 2 4; do timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 128 --spp 128 --count defaults lm=1 2>&1 | tee -a $OUT; done
timeout -k 10 300 python tools/sweep.py --scene c3n --depth 4 --frames 128 --spp 128 --count defaults lm=1 2>&1 | tee -a $OUT
