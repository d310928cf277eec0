#!/bin/bash
# round 5: full GPU suite + bench at the tree with the in-launch sort; any-hit A/B
OUT=gpurun_out/r05_17
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -4 $OUT/pytest.txt
for V in 1 0 1 0; do
  for A in "--config c5i --query any" "--config c4 --query any" "--config c2 --query any"; do
    python scripts/run_query.py $A --steps 200 --warmup 40 --opt sort_inline=$V 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('sort_inline=$V', r['config'], r['query'], r['rays'], r['tris'], r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab_any.txt
  done
done
cat $OUT/ab_any.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err
python -c "
import json
for f in ('bench','bench_driver_args'):
    r=json.loads(open('gpurun_out/r05_17/'+f+'.json').read().strip().splitlines()[-1]); print(f, r['value'], r['ms_per_step'], r.get('value_warmup_requested'), r['roofline']['kernel_avg_ms'], r['roofline']['frac'], r['verified'])
"
