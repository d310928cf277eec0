#!/bin/bash
# round 6, GPU job 17: what the float64 drain costs the COUNT / LOCATION launches (leaf phases by wave vote: many lanes test
# at once, so some lane parks a test in a fifth of the leaf steps): noexact = undecided tests count as misses (timing only:
# a count launch culls nothing, so the traversal is the same) against the shipped library
mkdir -p gpurun_out; OUT=gpurun_out/r06_count17.txt; : > $OUT
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base noexact base noexact; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c4 --query count --steps 30 --warmup 20
  TAG=$V; Q --config c5i --query count --steps 30 --warmup 20
  TAG=$V; Q --config c4 --query location --steps 20 --warmup 10
  TAG=$V; Q --config c5s --query count --steps 8
done
cat $OUT
