#!/bin/bash
# round 5, fourth GPU run: the suite on the tree with the new bench.py / sharded.py; the bench line; N-rank functional runs over gloo
OUT=gpurun_out/r05_4
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err; head -c 400 $OUT/bench_driver_args.json; echo
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 400 $OUT/bench_default.json; echo
for A in "" "--scaling strong" "--workload c5ii --total-rays 4000000"; do
  python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --min-warmup-ms 0 --no-cpu-baseline $A > $OUT/gloo2_$(echo $A | tr -d ' -').json 2>> $OUT/gloo2.err
  tail -c 300 $OUT/gloo2_$(echo $A | tr -d ' -').json; echo
done
TRIRO_PREFLIGHT_FAIL=slot,packed python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --min-warmup-ms 0 --no-cpu-baseline > $OUT/gloo2_fail2.json 2>> $OUT/gloo2.err
python bench.py --gpus 8 --backend gloo --steps 4 --warmup 2 --min-warmup-ms 0 --no-cpu-baseline > $OUT/gloo8.json 2>> $OUT/gloo2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_4/gloo*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, r['value'], r['verified'], r['config']['exchange_mode_used'], {k:(v.get('value'), v.get('verified'), v.get('error')) for k,v in r.items() if isinstance(v,dict) and ('rays_total' in v or 'error' in v)})
    except Exception as e: print(f, 'ERR', e)
PY
