#!/bin/bash
# The numbers DESIGN.md quotes, in one GPU call: every BASELINE config in the default (bit-exact) and the local-majorant mode,
# depths, protocol variants, ray caster.  usage: tools/final_numbers.sh <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
OUT=gpurun_out/${1:-final}_numbers.log
: > $OUT
for sc in c2 c3 c3n c4 c5; do timeout -k 10 300 python tools/sweep.py --scene $sc --frames 256 --spp 256 --count defaults lm=1 2>&1 | tee -a $OUT; done
for d in 2 3 4 6; do timeout -k 10 300 python tools/sweep.py --scene c3 --depth $d --frames 256 --spp 256 defaults split=0 split=2 queue=0 lm=1 2>&1 | tee -a $OUT; done
for d in 2 4; do timeout -k 10 300 python tools/sweep.py --scene c3n --depth $d --frames 128 --spp 128 defaults lm=1 2>&1 | tee -a $OUT; done
timeout -k 10 300 python tools/sweep.py --scene c5 --depth 2 --frames 128 --spp 128 defaults lm=1 2>&1 | tee -a $OUT
timeout -k 10 300 python tools/sweep.py --scene c3b --depth 1 --frames 256 --spp 256 defaults lm=1 2>&1 | tee -a $OUT
timeout -k 10 300 python tools/sweep.py --scene c3 --frames 256 --spp 256 defaults fast_math=1 queue=0 empty_skip=0 2>&1 | tee -a $OUT
timeout -k 10 300 python tools/sweep.py --scene c3 --depth 3 --frames 128 --spp 128 defaults env_nee=1 2>&1 | tee -a $OUT
for sc in c3 c3n c5; do timeout -k 10 300 python tools/per_frame.py --scene $sc 2>&1 | tee -a $OUT; done
for d in 2 4; do timeout -k 10 300 python tools/per_frame.py --scene c3 --depth $d 2>&1 | tee -a $OUT; done
timeout -k 10 300 python tools/per_frame.py --scene c3 frame_ahead=0 2>&1 | tee -a $OUT
timeout -k 10 300 python tools/sweep.py --scene c3 --frames 256 --spp 256 --shard 16,3,8 defaults 2>&1 | tee -a $OUT
timeout -k 10 300 python tools/raycast_time.py 2>&1 | tee -a $OUT
