#!/bin/bash
# round 4, GPU call 3: wide-node parity + A/B, expansion kernels with 96-bit row gathers
OUT=gpurun_out/r04_run3
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q -p no:cacheprovider > $OUT/pytest_wide.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_wide.txt
tail -15 $OUT/pytest_wide.txt
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -p no:cacheprovider -k "expansion" > $OUT/pytest_expand.txt 2>&1
tail -3 $OUT/pytest_expand.txt
timeout 900 python scripts/round4/ab_wide.py > $OUT/ab_wide.jsonl 2> $OUT/ab_wide.err
tail -3 $OUT/ab_wide.err
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run3/ab_wide.jsonl"):
    r = json.loads(ln)
    print(r["config"][:34], r["query"], "wide", r["wide"], r["ms"], "ms", r["grays_per_s"], "Grays/s", r["identical_to_binary"])
PY
E="python bench.py --steps 300 --warmup 30"
for m in 0 1 3; do
    timeout 600 $E --emulate-world 8 --opt expand4=$m --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
timeout 600 $E --emulate-world 8 --opt expand4=1 --arrival copy >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share auto --chunks 1 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.35 --chunks 1 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:22], c.get('opts'), 'share', c['dst_share'] and round(c['dst_share'],2), 'ch', c['chunks'], 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
