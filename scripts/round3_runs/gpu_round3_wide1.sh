#!/bin/bash
# two-level records (option wide_nodes) under the streaming launch: parity tests, then A/B on the incoherent configs
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | tail -8
for R in 1 2; do
for O in "" "--opt wide_nodes=1"; do
  for A in "--config c5s --query closest --steps 10 --warmup 4" "--config c3 --query closest --steps 10 --warmup 4" "--config c3 --query any --steps 10 --warmup 4" "--config c3 --query count --steps 6 --warmup 3" "--config c5i --query closest --res 2048 --opt stream=2 --steps 20 --warmup 8" "--config c5i --query closest --opt stream=2 --steps 40 --warmup 10"; do
    timeout 120 python scripts/run_query.py $A $O 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('wide' if '$O' else 'base', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
done
