#!/bin/bash
# round 6, GPU job 31: the unordered two-phase kernels (the multi-hit list query) at seven waves per SIMD (72 registers, 2 spills)
mkdir -p gpurun_out; OUT=gpurun_out/r06_u7_31.txt; : > $OUT
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base u7 base u7; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c4 --query location --steps 20 --warmup 10
  TAG=$V; Q --config c5i --query location --steps 20 --warmup 10
  TAG=$V; Q --config c2 --query location --steps 20 --warmup 10
  TAG=$V; Q --config terrain --query location --steps 20 --warmup 10
  TAG=$V; Q --config room --query location --steps 20 --warmup 10
  TAG=$V; Q --config soup --query location --steps 8
done
cat $OUT
