#!/bin/bash
# small launches split half of their blocks while a fifth of the wave slots stays free: new build vs split=2 (the old rule) forced
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "split or steal or adaptive" 2>&1 | tail -2
for R in 1 2; do
for O in "" "--opt split=2"; do
  for A in "--config c5i --query closest --res 256 --steps 100 --warmup 40" "--config c5i --query closest --res 320 --steps 100 --warmup 40" "--config c5i --query closest --res 384 --steps 100 --warmup 40" "--config c5i --query closest --res 448 --steps 100 --warmup 40" "--config c5i --query closest --res 512 --steps 100 --warmup 40" "--config terrain --query closest --res 512 --steps 100 --warmup 40" "--config terrain --query closest --res 640 --steps 100 --warmup 40" "--config c4 --query closest --res 384 --steps 100 --warmup 40" "--config c2 --query closest --res 384 --steps 100 --warmup 40" "--config room --query closest --res 512 --steps 100 --warmup 40" "--config c5i --query any --res 384 --steps 60 --warmup 30" "--config c5i --query count --res 384 --steps 40 --warmup 20"; do
    timeout 90 python scripts/run_query.py $A $O 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('new' if not '$O' else 'old', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
done
