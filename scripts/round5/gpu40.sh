#!/bin/bash
# round 5: is STEAL_MAX_RAYS (4 M) still the right boundary after the fused box test and the in-launch sort?  2048^2 and 2896^2 images of the headline mesh,
# automatic policy against stealing forced on / off
OUT=gpurun_out/r05_40; mkdir -p $OUT; : > $OUT/ab.txt
for RES in 2048 2896 4096; do
  for O in "" "--opt steal=64" "--opt steal=0" "--opt steal=64 --opt split=0"; do
    python scripts/run_query.py --config c5i --query closest --res $RES --steps 30 --warmup 12 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5i', r['rays'], 'closest', '$O' or 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
