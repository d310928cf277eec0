#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_interior.py -x -q -m gpu 2>&1 | tail -3
Q="timeout 90 python scripts/run_query.py --steps 60 --warmup 30"
(
for L in 0 1 0 1; do
for A in "--config c5i --query closest" "--config c5i --query closest --res 512" "--config c5i --query closest --res 2048" "--config c2 --query closest" "--config c4 --query closest" "--config c4 --query count" "--config room --query closest" "--config c5i --query count" "--config c3 --query any --steps 10 --warmup 4" "--config c3 --query closest --steps 10 --warmup 4" "--config c5s --query closest --steps 10 --warmup 4"; do
  $Q $A --opt node_layout=$L
done; done
) 2>&1 | grep -v amdgpu > gpurun_out/r3w_layout_ab.jsonl
python3 -c "
import json
for ln in open('gpurun_out/r3w_layout_ab.jsonl'):
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['query'], r['rays'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])
"
timeout 200 python scripts/bench_build.py 2>/dev/null | cut -c1-200
