#!/bin/bash
# round 4, GPU call 1: the new multi-rank-on-one-GPU tests, the destination-rank emulation, the bench line
OUT=gpurun_out/r04_run1
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_round4.py -x -q -p no:cacheprovider > $OUT/pytest_round4.txt 2>&1
echo "pytest round4 rc=$?" >> $OUT/pytest_round4.txt
tail -5 $OUT/pytest_round4.txt
E="python bench.py --steps 300 --warmup 30"
for args in "--emulate-world 8" "--emulate-world 8 --opt expand4=0" "--emulate-world 8 --arrival-priority" "--emulate-world 4" "--emulate-world 2" \
            "--emulate-world 8 --scaling strong" "--emulate-world 8 --scaling strong --dst-share auto" \
            "--emulate-world 8 --workload c5ii" "--emulate-world 8 --workload c5ii --dst-share auto" \
            "--emulate-world 8 --workload c5ii --dst-share auto --chunks 1" "--emulate-world 8 --workload c5ii --dst-share 0.4" \
            "--emulate-world 8 --workload c5ii --dst-share auto --arrival-priority"; do
  echo "== $args" >> $OUT/emulate.txt
  timeout 600 $E $args >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:40], 'share', c['dst_share'], 'ch', c['chunks'], 'prio', c['arrival_priority'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'verified', r['verified'])
"
