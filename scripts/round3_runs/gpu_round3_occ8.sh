#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "lds_staged" 2>&1 | tail -3
timeout 900 python scripts/fuzz_parity.py --iters 80 --seed 5 2>&1 | grep "MISMATCH\|done"
Q="timeout 90 python scripts/run_query.py --query closest --steps 80 --warmup 40"
for R in 1 2; do
for A in "--config c5i" "--config c5i --res 512" "--config c5i --res 2048" "--config c5i --res 768" "--config c4 --opt grid_nodes=2" "--config c2 --opt grid_nodes=2" "--config room --opt grid_nodes=2"; do
  for O in 0 1; do
    $Q $A --opt occ8=$O 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['rays'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])"
  done
done; done
