#!/bin/bash
# round 5, first GPU contact of the fused conservative box test: the whole GPU suite, then fused vs contract form
OUT=gpurun_out/r05_1
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
rm -f $OUT/ab_fuse.txt
bash scripts/round5/ab.sh $OUT/ab_fuse.txt base nofuse
cat $OUT/ab_fuse.txt
