#!/bin/bash
# after the change of the anchoring condition (the whole box within tmax): GPU suite, bench line, far-camera timing
OUT=gpurun_out/r06_final5
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -3 $OUT/pytest_gpu_full.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 300 $OUT/bench_default.json; echo
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/far.txt; }
for far in 1 100 10000; do TAG="far=$far"; Q --config c5i --query closest --steps 40 --warmup 30 --far $far; done
cat $OUT/far.txt
timeout 600 python scripts/fuzz_parity.py --iters 200 --seed 631 > $OUT/fuzz_seed631.txt 2>&1; tail -1 $OUT/fuzz_seed631.txt
