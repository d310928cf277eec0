#!/bin/bash
# round 5: last tree: the GPU suite, the bench lines, count auto at 12.5 M / 25 M
OUT=gpurun_out/r05_60; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err
python -c "
import json
for f in ('bench','bench_driver_args'):
    r=json.loads(open('gpurun_out/r05_60/'+f+'.json').read().strip().splitlines()[-1]); rl=r['roofline']
    print(f, r['value'], r['ms_per_step'], r.get('value_warmup_requested'), rl['kernel_avg_ms'], rl['kernel_median_ms'], rl['kernel_min_ms'], rl['frac'], r['verified'])
"
for N in 12500000 25000000; do python scripts/run_query.py --config c5s --query count --rays $N --steps 10 --warmup 4 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5s', r['rays'], 'count auto', r['ms_mean'], r['ms_min'])"; done
PROFILE_ROUND=r05 timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; wc -l $OUT/configs.jsonl
