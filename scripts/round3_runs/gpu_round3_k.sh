#!/bin/bash
mkdir -p gpurun_out
Q="timeout 90 python scripts/run_query.py --steps 40 --warmup 20 --query count"
(
for CFG in "c4" "room --res 1280" "c2"; do
  $Q --config $CFG
  $Q --config $CFG --opt split_outlier=0 --opt usteal_tail=0 --opt split=4
  $Q --config $CFG --opt split_outlier=0 --opt usteal_tail=0 --opt split=3
  $Q --config $CFG --opt split_outlier=0 --opt usteal_tail=0 --opt split=2
  $Q --config $CFG --opt split_outlier=0 --opt usteal_tail=0 --opt split=2 --opt split_steal=2
  $Q --config $CFG --opt split_outlier=4 --opt usteal_tail=0 --opt split=2 --opt split_steal=4
done
) > gpurun_out/r3k_usteal2.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r3k_usteal2.jsonl | python3 -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['rays'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])
"
