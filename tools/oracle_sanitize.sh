#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU oracle (sanitizers are CPU-only on this pool): builds an
# instrumented oracle, swaps it in for the CPU test suite's oracle-facing tests, restores the regular build.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/oracle
cp libsvr_oracle.so /tmp/libsvr_oracle_regular.so
gcc -O1 -g -std=c11 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -mfma -fsanitize=address,undefined -fno-omit-frame-pointer \
    svr_oracle.c svr_io_oracle.c -o libsvr_oracle.so -shared -fopenmp -lm || { cp /tmp/libsvr_oracle_regular.so libsvr_oracle.so; exit 1; }
cd $ROOT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_oracle_kat.py tests/test_io_cpu.py tests/test_host.py -q -p no:cacheprovider 2>&1 | tail -15
rc=${PIPESTATUS[0]}
cp /tmp/libsvr_oracle_regular.so oracle/libsvr_oracle.so
touch oracle/libsvr_oracle.so
exit $rc
