#!/bin/bash
# round 4, GPU call 2: expansion kernel variants x arrival model, expansion overlapping the trace
OUT=gpurun_out/r04_run2
mkdir -p $OUT
python -m pytest tests/test_gpu_round4.py -x -q -p no:cacheprovider -k "expansion or emulated or async" > $OUT/pytest_round4.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_round4.txt
tail -3 $OUT/pytest_round4.txt
E="python bench.py --steps 300 --warmup 30"
for m in 0 1 2 3; do
  for arr in copy none; do
    timeout 600 $E --emulate-world 8 --opt expand4=$m --arrival $arr >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
  done
done
for m in 1 3; do
  timeout 600 $E --emulate-world 8 --workload c5ii --dst-share auto --chunks 1 --opt expand4=$m --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
  timeout 600 $E --emulate-world 8 --workload c5ii --dst-share auto --opt expand4=$m --arrival copy >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.3 --opt expand4=3 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:22], 'share', c['dst_share'] and round(c['dst_share'],2), 'ch', c['chunks'], 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
