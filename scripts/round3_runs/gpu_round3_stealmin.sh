#!/bin/bash
# trip from which the rays of an ordinary (unsplit) block give subtrees away: default 64 (option steal = 1) against 48 ... 8
cd $GRAFT_REPO_ROOT
for S in 1 48 32 24 16 8 1; do
  for A in "--config c5i --query closest --res 512 --steps 100 --warmup 40" "--config c5i --query closest --res 768 --steps 100 --warmup 40" "--config c5i --query closest --steps 100 --warmup 40" "--config c4 --query closest --res 512 --steps 100 --warmup 40" "--config c4 --query closest --steps 100 --warmup 40" "--config c2 --query closest --steps 100 --warmup 40" "--config room --query closest --steps 100 --warmup 40" "--config c5i --query any --res 512 --steps 60 --warmup 30"; do
    timeout 90 python scripts/run_query.py $A --opt steal=$S 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('steal=$S', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
