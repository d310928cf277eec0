#!/bin/bash
# round 5: binary (wide=0) against 8-wide nodes (wide=1) in the streaming launch, per QUERY, on the headline mesh (1.31 M triangles) and at 5.2 M triangles
OUT=gpurun_out/r05_45; mkdir -p $OUT; : > $OUT/ab.txt
for rep in 1 2; do for SD in 8 9; do for Q in any closest first count; do for N in 4194304 12500000; do for O in "--opt wide=0" "--opt wide=1"; do
  [ $Q = count ] && [ $N = 4194304 ] && continue
  python scripts/run_query.py --config c5s --subdiv $SD --query $Q --rays $N --steps 12 --warmup 5 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5s', r['tris'], r['rays'], '$Q', '$O', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
done; done; done; done; done
sort -k2,4 -s $OUT/ab.txt
