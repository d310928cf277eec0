#!/bin/bash
mkdir -p gpurun_out
Q="python scripts/run_query.py --steps 40 --warmup 20 --query closest"
(
for CFG in "room" "room --res 1280" "room --res 2560" "c5i --res 512" "c5i --res 768" "c5i --res 1024" "c4 --res 512" "c4" "c2 --res 512" "c2"; do
  for OUT in 0 8 12 16 24; do
    $Q --config $CFG --opt split_outlier=$OUT
  done
  $Q --config $CFG --opt split_outlier=12 --opt split_floor=0
  $Q --config $CFG --opt split_outlier=16 --opt split_floor=80
done
) > gpurun_out/r3f_outlier.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r3f_outlier.jsonl | python3 -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['rays'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])
"
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "split" 2>&1 | tail -3
