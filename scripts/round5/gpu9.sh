#!/bin/bash
# round 5: inside test first in the full predicate (TR_MT_FIRST): parity on the suites that stress the leaf tests, then A/B against the tree without it
OUT=gpurun_out/r05_9
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round5.py tests/test_gpu_soup.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -4 $OUT/pytest.txt
rm -f $OUT/ab_mt.txt
bash scripts/round5/ab.sh $OUT/ab_mt.txt base mt0
cat $OUT/ab_mt.txt
