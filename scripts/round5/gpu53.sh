#!/bin/bash
# round 5: streaming from 3 M (small meshes) / above 4 M rays (others) after the adaptive stealing: full suite, the automatic policy, the c5ii emulation
OUT=gpurun_out/r05_53; mkdir -p $OUT; : > $OUT/auto.txt
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.txt
for C in c3 c5s; do for Q in closest any first count; do for N in 1048576 2200000 3000000 4194304 6000000 12500000; do
  python scripts/run_query.py --config $C --query $Q --rays $N --steps 16 --warmup 8 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$C', r['rays'], '$Q', 'auto', r['ms_mean'], r['ms_min'])" >> $OUT/auto.txt
done; done; done
cat $OUT/auto.txt
for extra in "--workload c5ii --steps 10 --warmup 3" "--workload c5ii --steps 10 --warmup 3 --arrival copy"; do
  arr="--arrival none"; case "$extra" in *arrival*) arr="";; esac
  timeout 600 python bench.py --emulate-world 8 $arr --records slot --steps 200 --warmup 50 $extra 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=r['emulation']; c=r['config']; print('emulate c5ii', c['dst_share'], c['arrival'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'implied', e['implied_scaling_vs_1gpu'], r['verified'])"
done
