#!/bin/bash
# round 4, GPU call 6: slot-form records; full round-4 + wide tests; wide streaming on coherent configs; f3 PMC evidence
OUT=gpurun_out/r04_run6
mkdir -p $OUT
REPO=$PWD
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_wide.py -x -q -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -5 $OUT/pytest.txt
timeout 600 python scripts/round4/expand_micro.py --big 0 > $OUT/expand_micro.jsonl 2> $OUT/expand_micro.err
grep -i "slot\|shard\|image records (55" $OUT/expand_micro.jsonl
E="python bench.py --steps 300 --warmup 30"
timeout 600 $E --emulate-world 8 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival copy >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
TRIRO_PACKED_SLOTS=0 timeout 600 $E --emulate-world 8 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --arrival none --opt expand_cus=2 >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 4 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 2 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
for sh in auto 0.35 0.2; do
  timeout 600 $E --emulate-world 8 --workload c5ii --dst-share $sh --chunks 1 --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.35 --arrival copy >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
timeout 600 $E --emulate-world 8 --scaling strong --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:22], 'N', r['emulated_world'], c.get('opts'), c.get('record_form'), 'share', c['dst_share'] and round(c['dst_share'],2), 'ch', c['chunks'], 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
# wide streaming on the coherent configs (is an 8-wide walk worth building for the direct launch?)
for a in "--config c5i --query closest" "--config c5i --query closest --opt stream=2 --opt wide=0" "--config c5i --query closest --opt stream=2 --opt wide=1" \
         "--config c4 --query count" "--config c4 --query count --opt stream=2 --opt wide=0" "--config c4 --query count --opt stream=2 --opt wide=1" \
         "--config c4 --query closest" "--config c4 --query closest --opt stream=2 --opt wide=1" "--config c2 --query closest" "--config c2 --query closest --opt stream=2 --opt wide=1"; do
  echo "== $a" >> $OUT/wide_coherent.txt
  timeout 300 python scripts/run_query.py $a --steps 30 --warmup 12 >> $OUT/wide_coherent.txt 2>&1
done
grep -v "^\[" $OUT/wide_coherent.txt | cut -c1-200
# f3 evidence: the streaming launch on a C5(ii) shard, binary grid nodes vs 8-wide nodes, kernel trace + PMC passes
KERNEL_KEY=k_query_stream bash scripts/profile_query.sh r04_c5s_bin --config c5s --query closest --opt wide=0 > $OUT/prof_bin.txt 2>&1
KERNEL_KEY=k_query_wide bash scripts/profile_query.sh r04_c5s_wide --config c5s --query closest --opt wide=1 > $OUT/prof_wide.txt 2>&1
tail -25 $OUT/prof_bin.txt; tail -25 $OUT/prof_wide.txt
