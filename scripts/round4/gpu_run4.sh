#!/bin/bash
# round 4, GPU call 4: fused wide kernel (variants), buffer-load expansion + grid cap
OUT=gpurun_out/r04_run4
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q -p no:cacheprovider > $OUT/pytest_wide.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_wide.txt
tail -5 $OUT/pytest_wide.txt
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -p no:cacheprovider -k "expansion" > $OUT/pytest_expand.txt 2>&1
tail -3 $OUT/pytest_expand.txt
for var in default w6 late late6; do
  if [ $var = default ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$var/libtriro_hip.so; fi
  echo "== $var" >> $OUT/ab_wide.txt
  AB_WIDE_BIG=$([ $var = default ] && echo 1 || echo 0) timeout 900 python scripts/round4/ab_wide.py > $OUT/ab_wide_$var.jsonl 2>> $OUT/ab_wide.txt
  python - $OUT/ab_wide_$var.jsonl $var <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    print(sys.argv[2], r["config"][:30], r["query"], "wide", r["wide"], r["ms"], "ms", r["grays_per_s"], "Grays/s", r["identical_to_binary"], r.get("stats"))
PY
done
unset TRIRO_HIP_LIBRARY
E="python bench.py --steps 300 --warmup 30"
for o in "expand4=0" "expand4=1" "expand4=1 --opt expand_cus=1" "expand4=1 --opt expand_cus=2" "expand4=1 --opt expand_cus=4" "expand4=1 --opt expand_cus=8"; do
    timeout 600 $E --emulate-world 8 --opt $o --arrival none >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
timeout 600 $E --emulate-world 8 --opt expand4=1 --opt expand_cus=2 --arrival copy >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
for o in "expand4=1" "expand4=1 --opt expand_cus=2" "expand4=1 --opt expand_cus=4"; do
  timeout 600 $E --emulate-world 8 --workload c5ii --dst-share 0.35 --chunks 1 --arrival none --opt $o >> $OUT/emulate.jsonl 2>> $OUT/emulate.txt
done
cat $OUT/emulate.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r = json.loads(ln); e = r['emulation']; c = r['config']
    print(c['workload'][:22], c.get('opts'), 'share', c['dst_share'] and round(c['dst_share'],2), 'ch', c['chunks'], 'arr', c['arrival'], '| plain', e['plain_1gpu_ms_per_step'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], e['expansion_GBps'], 'GB/s | implied', e['implied_scaling_vs_1gpu'], 'ok', r['verified'])
"
