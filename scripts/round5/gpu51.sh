#!/bin/bash
# round 5: the DIRECT launch on incoherent flat batches below the streaming boundary (1 M hash rays): automatic policy against forced settings
OUT=gpurun_out/r05_51; mkdir -p $OUT; : > $OUT/ab.txt
for C in c3 c5s; do for Q in closest any count; do
  for O in "" "--opt adaptive=0" "--opt split=0" "--opt steal=0" "--opt steal=16" "--opt steal=32" "--opt xcd_chunk=0" "--opt usteal=0" "--opt split=3"; do
    python scripts/run_query.py --config $C --query $Q --rays 1048576 --steps 60 --warmup 20 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$C', r['rays'], '$Q', '$O' or 'auto', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
  done
done; done
cat $OUT/ab.txt
