#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (GPU sanitizers are not available on this pool):
#  1. tests/host_sim -- the product's own per-lane state machine and box tests (csrc/tr_math.h, tr_bvh.h, tr_lbvh.h, tr_wide.h)
#     compiled with g++ -- under tests/test_host_sim.py
#  2. the oracle (oracle/triro_oracle.c) under tests/test_oracle.py and tests/test_geometry_f64.py
# The instrumented libraries replace the ordinary ones for the run and are removed / restored afterwards.
set -u
cd "$(dirname "$0")/../.."
SAN="-fsanitize=undefined,address -fno-sanitize-recover=undefined"
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:log_path=/tmp/triro_asan UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/triro_ubsan
rm -f /tmp/triro_asan* /tmp/triro_ubsan*
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -mfma -Wno-unknown-pragmas $SAN -o tests/host_sim/libhost_sim.so tests/host_sim/host_sim.cpp || exit 1
touch tests/host_sim/libhost_sim.so
LD_PRELOAD="$PRE" timeout 2400 python -m pytest tests/test_host_sim.py -q -p no:cacheprovider; rc1=$?
rm -f tests/host_sim/libhost_sim.so                       # (rebuilt without instrumentation by the next test run)
cp oracle/libtriro_oracle.so /tmp/libtriro_oracle_plain.so
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off -mfma -fopenmp $SAN -shared -o oracle/libtriro_oracle.so oracle/triro_oracle.c -lm || exit 1
LD_PRELOAD="$PRE" OMP_NUM_THREADS=4 timeout 2400 python -m pytest tests/test_oracle.py tests/test_geometry_f64.py -q -p no:cacheprovider; rc2=$?
cp /tmp/libtriro_oracle_plain.so oracle/libtriro_oracle.so
echo "host_sim rc=$rc1 oracle rc=$rc2; sanitizer reports:"; ls /tmp/triro_asan* /tmp/triro_ubsan* 2>/dev/null || echo "  none"
exit $(( rc1 | rc2 ))
