#!/bin/bash
mkdir -p gpurun_out
Q="python scripts/run_query.py --steps 40 --warmup 16 --query closest"
(
for CFG in "room" "room --res 1280" "room --res 2560" "c5i --res 512" "c5i --res 1024" "c4 --res 512" "c4" "c2 --res 512" "c2"; do
  $Q --config $CFG
  $Q --config $CFG --opt split=0
  $Q --config $CFG --opt split=0 --opt tile_small=0 --opt tile=0
  $Q --config $CFG --opt split=0 --opt tile_small=0 --opt tile=0 --opt steal=0
  $Q --config $CFG --opt block_size=64
  $Q --config $CFG --opt split=0 --opt tile_small=3
  $Q --config $CFG --opt split=0 --opt tile_small=3 --opt steal=0
done
) > gpurun_out/r3e_policy.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r3e_policy.jsonl | python3 -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['rays'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])
"
