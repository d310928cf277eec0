#!/bin/bash
# round 5: the pruned tree (107 kernels, no node-flavour tuner): the whole suite, a fuzz run, steady-state figures, any-hit on grid nodes
OUT=gpurun_out/r05_7
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
timeout 600 python scripts/fuzz_parity.py --iters 150 --seed 503 > $OUT/fuzz_503.txt 2>&1; tail -1 $OUT/fuzz_503.txt
rm -f $OUT/ab.txt
bash scripts/round5/ab.sh $OUT/ab.txt base
cat $OUT/ab.txt
for G in 1 2; do for C in c5i c4 c2 terrain room; do
  python scripts/run_query.py --config $C --query any --steps 60 --warmup 40 --opt grid_nodes=$G 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])" >> $OUT/any_flavour.txt
done; done
cat $OUT/any_flavour.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('value','value_warmup_requested')}, {k:v for k,v in r['roofline'].items() if 'ms' in k})"
python scripts/bench_build.py 2>/dev/null | tail -3
