#!/bin/bash
# round 5: last checks at the last tree: 400 fuzz iterations, the bench lines
OUT=gpurun_out/r05_50; mkdir -p $OUT
for S in 541 542; do timeout 1500 python scripts/fuzz_parity.py --iters 200 --seed $S > $OUT/fuzz_seed$S.txt 2>&1; echo "seed $S rc=$?"; tail -1 $OUT/fuzz_seed$S.txt; done
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench.err
python -c "
import json
for f in ('bench','bench_driver_args'):
    r=json.loads(open('gpurun_out/r05_50/'+f+'.json').read().strip().splitlines()[-1]); rl=r['roofline']
    print(f, r['value'], r['ms_per_step'], r.get('value_warmup_requested'), rl['kernel_avg_ms'], rl['kernel_median_ms'], rl['kernel_min_ms'], rl['frac'], r['verified'])
"
