#!/bin/bash
# round 5: is STREAM_MIN_RAYS (2 M) still the right boundary?  Incoherent hash rays on the C2 mesh (binary streaming launch) and the headline mesh
# (8-wide streaming launch): automatic policy (direct below 2 M, probe + both shapes from 2 M on) against the streaming launch forced (stream=2) and
# the direct launch forced (stream=0).  Second pass: the sizes around the crossovers.
OUT=gpurun_out/r05_41; mkdir -p $OUT; : > $OUT/ab2.txt
for Q in closest any count; do
  for N in 1200000 1500000; do for O in "--opt stream=2" "--opt stream=0"; do
    python scripts/run_query.py --config c3 --query $Q --rays $N --steps 40 --warmup 12 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c3', r['rays'], '$Q', '$O' or 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab2.txt
  done; done
  for N in 2600000 3000000 3500000; do for O in "--opt stream=2" "--opt stream=0"; do
    python scripts/run_query.py --config c5s --query $Q --rays $N --steps 40 --warmup 12 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5s', r['rays'], '$Q', '$O' or 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab2.txt
  done; done
done
for N in 2200000 3000000 4194304; do for O in "--opt stream=2" "--opt stream=0"; do
  python scripts/run_query.py --config c5s --subdiv 9 --query closest --rays $N --steps 30 --warmup 10 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5s-5.2M-tris', r['rays'], 'closest', '$O' or 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab2.txt
done; done
cat $OUT/ab2.txt
