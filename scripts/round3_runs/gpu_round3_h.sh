#!/bin/bash
# every command under its own timeout: a hung kernel must not eat the GPU budget
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "unordered_count" 2>&1 | tail -12
Q="timeout 90 python scripts/run_query.py --steps 40 --warmup 20 --query count"
(
for CFG in "c4" "c4 --res 512" "c5i" "c2" "room" "room --res 1280"; do
  $Q --config $CFG --opt usteal=0
  $Q --config $CFG
  $Q --config $CFG --opt usteal_tail=0
  $Q --config $CFG --opt usteal_tail=64
  $Q --config $CFG --opt usteal_tail=128
  $Q --config $CFG --opt usteal_tail=192
  $Q --config $CFG --opt usteal=4
  $Q --config $CFG --opt usteal=64
  $Q --config $CFG --opt split_steal=2
  $Q --config $CFG --opt split_floor=0
  $Q --config $CFG --opt split=0 --opt usteal_tail=0
done
$Q --config c3 --opt usteal=0
$Q --config c3
) > gpurun_out/r3h_usteal.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r3h_usteal.jsonl | python3 -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['rays'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])
"
