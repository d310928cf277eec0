#!/bin/bash
# round 5: refill threshold of the streaming launches (binary and 8-wide nodes) after the fused box test
OUT=gpurun_out/r05_11
mkdir -p $OUT
for R in 32 8 12 16 20 24 28 32; do
  for C in "c5s --query closest" "c5s --query closest --opt wide=0" "c5s --query count" "c5s --query any" "c3 --query any" "c3 --query closest" "c5s --query closest --subdiv 9"; do
    python scripts/run_query.py --config $C --steps 12 --warmup 6 --opt stream_refill=$R 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])" >> $OUT/refill.txt
  done
done
cat $OUT/refill.txt
