#!/bin/bash
OUT=gpurun_out/r04_run14
mkdir -p $OUT
E="python bench.py --steps 300 --warmup 30"
for i in 1 2; do
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.35 --chunks 1 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.2 --chunks 1 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.35 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:30], 'N', r['emulated_world'], 'share', c['dst_share'], 'ch', c['chunks'], 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
