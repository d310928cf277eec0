#!/bin/bash
OUT=gpurun_out/r04_run22
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -15 $OUT/pytest_gpu_full.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 300 $OUT/bench_default.json; echo
timeout 600 python scripts/round4/exp_order_transfer.py > $OUT/order_transfer.json 2>/dev/null; cat $OUT/order_transfer.json
