#!/bin/bash
# full GPU suite + fuzz with the 4-byte records
OUT=gpurun_out/r04_run29
mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -5 $OUT/pytest_gpu_full.txt
for seed in 51 52; do
  timeout 900 python scripts/fuzz_parity.py --iters 120 --seed $seed > $OUT/fuzz_seed$seed.txt 2>&1; tail -1 $OUT/fuzz_seed$seed.txt
done
