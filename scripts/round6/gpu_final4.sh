#!/bin/bash
# after the clean-up of dead A/B variants: the GPU suite twice (flakiness), smoke, the default bench line, two more fuzz seeds
OUT=gpurun_out/r06_final4
mkdir -p $OUT
for k in 1 2; do
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full_$k.txt 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_gpu_full_$k.txt; tail -2 $OUT/pytest_gpu_full_$k.txt
done
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt | cut -c1-100
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 300 $OUT/bench_default.json; echo
for S in 621 622; do
  timeout 1200 python scripts/fuzz_parity.py --iters 200 --seed $S > $OUT/fuzz_seed$S.txt 2>&1; tail -1 $OUT/fuzz_seed$S.txt
done
