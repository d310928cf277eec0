#!/bin/bash
# round 5: the opt-in maximum-size case at the final tree (21 M and 84 M triangles: arrays above 4 GiB, 64-bit addressing)
OUT=gpurun_out/r05_27; mkdir -p $OUT
TRIRO_TEST_HUGE=${HUGE:-1} timeout 1500 python -m pytest tests/test_gpu_round2.py -q -p no:cacheprovider -k large_meshes > $OUT/pytest_huge.txt 2>&1
echo "rc=$?" >> $OUT/pytest_huge.txt; tail -3 $OUT/pytest_huge.txt
