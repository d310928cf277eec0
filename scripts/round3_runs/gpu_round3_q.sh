#!/bin/bash
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "unordered_count" 2>&1 | tail -8
timeout 300 python -m pytest tests/test_gpu_interior.py tests/test_gpu_configs.py -x -q -m gpu -k "room or c4" 2>&1 | tail -4
Q="timeout 90 python scripts/run_query.py --steps 30 --warmup 16 --query location"
(
for CFG in "c4" "c4 --res 512" "c5i" "c2" "room" "room --res 1280"; do
  $Q --config $CFG --opt usteal=0
  $Q --config $CFG
done
) 2>&1 | grep -v amdgpu > gpurun_out/r3q_location.jsonl
python3 -c "
import json
for ln in open('gpurun_out/r3q_location.jsonl'):
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['rays'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])
"
