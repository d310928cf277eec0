#!/bin/bash
# round 5: the count launch's stealing threshold (16) on incoherent batches and on images: usteal = 4 / 8 / 16 (default) / 32
OUT=gpurun_out/r05_58; mkdir -p $OUT; : > $OUT/ab.txt
for U in 1 4 8 32; do
  for A in "--config c3 --rays 1048576" "--config c3 --rays 4194304" "--config c5s --rays 1048576" "--config c5s --rays 4194304" "--config c4" "--config c5i" "--config terrain" "--config room"; do
    python scripts/run_query.py $A --query count --steps 40 --warmup 15 --opt usteal=$U 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('usteal=$U', r['config'], r['rays'], 'count', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
  done
done
sort -k2,3 -s $OUT/ab.txt
