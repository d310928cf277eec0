#!/bin/bash
# round 5: the streaming boundary for COUNT (the direct count launch steals and splits; streaming 3.5 M incoherent rays was 48 % slower)
OUT=gpurun_out/r05_42; mkdir -p $OUT; : > $OUT/ab3.txt
for N in 2200000 4194304 6000000 10000000; do for O in "--opt stream=2" "--opt stream=0"; do
  python scripts/run_query.py --config c3 --query count --rays $N --steps 20 --warmup 8 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c3', r['rays'], 'count', '$O' or 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab3.txt
done; done
for N in 4194304 6000000 8000000 12500000 25000000; do for O in "--opt stream=2" "--opt stream=0"; do
  python scripts/run_query.py --config c5s --query count --rays $N --steps 16 --warmup 6 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5s', r['rays'], 'count', '$O' or 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab3.txt
done; done
for N in 6000000 10000000; do for Q in closest any; do for O in "--opt stream=2" "--opt stream=0"; do
  python scripts/run_query.py --config c3 --query $Q --rays $N --steps 16 --warmup 6 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c3', r['rays'], '$Q', '$O' or 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab3.txt
done; done; done
cat $OUT/ab3.txt
