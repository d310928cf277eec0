#!/bin/bash
OUT=gpurun_out/r04_run37
mkdir -p $OUT
TRIRO_TEST_HUGE=1 timeout 1500 python -m pytest tests/test_gpu_round2.py -q -p no:cacheprovider -k "large_meshes" > $OUT/pytest_huge.txt 2>&1
echo "rc=$?" >> $OUT/pytest_huge.txt; tail -5 $OUT/pytest_huge.txt
timeout 900 python -m pytest tests/test_gpu_round4.py -q -p no:cacheprovider -k "two_ranks or bench_two or emulated" 2>&1 | tail -2
