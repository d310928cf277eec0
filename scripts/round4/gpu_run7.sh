#!/bin/bash
OUT=gpurun_out/r04_run7
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_wide.py -x -q -p no:cacheprovider > $OUT/pytest_wide.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_wide.txt; tail -12 $OUT/pytest_wide.txt
bash scripts/round4/ab_wide_direct.sh $OUT
