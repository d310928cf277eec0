#!/bin/bash
OUT=gpurun_out/r04_run9
mkdir -p $OUT
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_wide.py -x -q -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -5 $OUT/pytest.txt
timeout 600 python scripts/round4/expand_micro.py --big 0 > $OUT/expand_micro.jsonl 2> $OUT/expand_micro.err
grep -i "slot" $OUT/expand_micro.jsonl
E="python bench.py --steps 300 --warmup 30"
timeout 600 $E --emulate-world 8 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival none --opt expand_tiles=0 >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival none --opt expand_cus=2 >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival copy >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 4 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:22], 'N', r['emulated_world'], c.get('opts'), c.get('record_form'), 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
for c in room c5i c4; do bash scripts/policy_matrix.sh $c "1024" 2>&1 | grep -v closest > $OUT/policy_$c.txt; cat $OUT/policy_$c.txt; done
