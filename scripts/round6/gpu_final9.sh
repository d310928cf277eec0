#!/bin/bash
# count at seven waves shipped: GPU suite, fuzz, bench line, per-config table
OUT=gpurun_out/r06_final9
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
for S in 651 652; do timeout 900 python scripts/fuzz_parity.py --iters 200 --seed $S > $OUT/fuzz_seed$S.txt 2>&1; tail -1 $OUT/fuzz_seed$S.txt; done
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 300 $OUT/bench_default.json; echo
PROFILE_ROUND=r06 timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; wc -l $OUT/configs.jsonl
