#!/bin/bash
# round 5: TR_STEAL_EARLY = 4 + one streaming boundary (above 4 M rays; count 8 M): full suite, the image configs, the automatic policy on incoherent batches, 200 fuzz iterations
OUT=gpurun_out/r05_57; mkdir -p $OUT; : > $OUT/auto.txt
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.txt
python bench.py --steps 400 --warmup 50 --no-companions --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', r['value'], r['ms_per_step'], r['verified'])" >> $OUT/auto.txt
for A in "--config c2 --query closest" "--config c4 --query closest" "--config c5i --query any" "--config c5i --query first" "--config terrain --query closest" "--config room --query closest" "--config c5i --res 512 --query closest" "--config c5i --query closest --flat" "--config c4 --query count"; do
  python scripts/run_query.py $A --steps 100 --warmup 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('image', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])" >> $OUT/auto.txt
done
for C in c3 c5s; do for Q in closest any first; do for N in 262144 1048576 2200000 3000000 4194304 6000000; do
  python scripts/run_query.py --config $C --query $Q --rays $N --steps 20 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$C', r['rays'], '$Q', 'auto', r['ms_mean'], r['ms_min'])" >> $OUT/auto.txt
done; done; done
cat $OUT/auto.txt
timeout 1500 python scripts/fuzz_parity.py --iters 200 --seed 551 > $OUT/fuzz_seed551.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_seed551.txt
