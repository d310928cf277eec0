#!/bin/bash
mkdir -p gpurun_out
for k in 1 2 3 4 5; do
timeout 300 python bench.py --no-cpu-baseline --no-companions 2>/dev/null | python3 -c "
import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); rl=r['roofline']; print('run', $k, r['value'], rl['kernel_avg_ms'], rl['node_flavour'], r['verified'])"
done
for C in "c2" "c4" "room" "c5i --res 512" "c5i --res 2048"; do
 for k in 1 2; do
  timeout 90 python scripts/run_query.py --config $C --query closest --steps 300 --warmup 300 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['rays'], r['ms_mean'], r['ms_min'])"
 done
done
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -m gpu -k "steady_state or last_launch or moving" 2>&1 | tail -3
