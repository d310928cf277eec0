#!/bin/bash
# round 5: the bench lines at the last tree (default flags, the driver's arguments)
OUT=gpurun_out/r05_32; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err
python -c "
import json
for f in ('bench','bench_driver_args'):
    r=json.loads(open('gpurun_out/r05_32/'+f+'.json').read().strip().splitlines()[-1]); rl=r['roofline']
    print(f, r['value'], r['ms_per_step'], r.get('value_warmup_requested'), rl['kernel_avg_ms'], rl['kernel_median_ms'], rl['kernel_min_ms'], rl['frac'], r['verified'])
"
tail -2 $OUT/bench.err
