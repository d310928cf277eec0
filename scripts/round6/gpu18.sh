#!/bin/bash
# round 6, GPU job 18: parked leaf tests of the unordered schedule (count / location) drained in batches (TR_DRAIN_BATCH = 8:
# base) against after every leaf step (batch0 = the tree before), 4 and 64; parity of the count / location / contains paths first
mkdir -p gpurun_out; OUT=gpurun_out/r06_batch18.txt; : > $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "count or location or contains or soup or terrain or interior or c4 or fuzz or round3 or usteal" > gpurun_out/r06_gputest18.txt 2>&1; tail -3 gpurun_out/r06_gputest18.txt
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base batch0 batch4 batch64 base batch0; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c4 --query count --steps 30 --warmup 20
  TAG=$V; Q --config c5i --query count --steps 30 --warmup 20
  TAG=$V; Q --config c4 --query location --steps 20 --warmup 10
  TAG=$V; Q --config c5i --query location --steps 20 --warmup 10
  TAG=$V; Q --config terrain --query location --steps 20 --warmup 10
  TAG=$V; Q --config room --query count --steps 30 --warmup 20
  TAG=$V; Q --config c5s --query count --steps 8
done
cat $OUT
