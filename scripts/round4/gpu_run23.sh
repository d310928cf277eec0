#!/bin/bash
OUT=gpurun_out/r04_run23
mkdir -p $OUT
PROFILE_ROUND=r04 timeout 1500 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; tail -3 $OUT/configs.err
python -c "
import json
for ln in open('$OUT/configs.jsonl'):
    r = json.loads(ln); print(r['config'][:70], r['ms'], r.get('mrays_per_s'), r.get('frac'))
"
timeout 900 python bench.py --workload c5ii --steps 20 --warmup 3 > $OUT/bench_c5ii.json 2> $OUT/bench_c5ii.err; head -c 400 $OUT/bench_c5ii.json; echo
timeout 600 python scripts/bench_build.py > $OUT/build.jsonl 2>/dev/null; tail -4 $OUT/build.jsonl | cut -c1-300
