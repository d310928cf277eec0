#!/bin/bash
# round 6, GPU job 32: the direct launch on the 8-wide nodes (the list query on big meshes) at six waves per SIMD (80 registers, 9-17 spills, four scratch instructions inside the trip of the list kernel)
mkdir -p gpurun_out; OUT=gpurun_out/r06_wd6_32.txt; : > $OUT
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base wd6 base wd6; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c4 --query location --steps 20 --warmup 10
  TAG=$V; Q --config c5i --query location --steps 20 --warmup 10
  TAG=$V; Q --config c2 --query location --steps 20 --warmup 10
  TAG=$V; Q --config terrain --query location --steps 20 --warmup 10
  TAG=$V; Q --config room --query location --steps 20 --warmup 10
  TAG=$V; Q --config soup --query location --steps 8
done
cat $OUT
