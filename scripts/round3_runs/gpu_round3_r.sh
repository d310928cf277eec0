#!/bin/bash
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "lds_staged" 2>&1 | tail -8
Q="timeout 90 python scripts/run_query.py --steps 40 --warmup 20 --query closest"
(
for CFG in "c5i" "c5i --res 512" "c5i --res 2048" "c2" "c4" "room" "room --res 1280"; do
  $Q --config $CFG
  $Q --config $CFG --opt grid_nodes=2
  $Q --config $CFG --opt lds_top=1
  $Q --config $CFG --opt lds_top=2
  $Q --config $CFG --opt block_size=256
done
) 2>&1 | grep -v amdgpu > gpurun_out/r3r_lds_top.jsonl
python3 -c "
import json
for ln in open('gpurun_out/r3r_lds_top.jsonl'):
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['rays'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])
"
