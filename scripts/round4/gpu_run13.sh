#!/bin/bash
OUT=gpurun_out/r04_run13
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -p no:cacheprovider > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
E="python bench.py --steps 300 --warmup 30"
timeout 600 $E --emulate-world 8 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival copy >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 4 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 2 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival none --res 2048 >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --scaling strong --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.35 --chunks 1 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:30], 'N', r['emulated_world'], c.get('opts'), 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
