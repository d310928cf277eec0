#!/bin/bash
# round 6, GPU job 19: the streaming kernels for any / first at SIX waves per SIMD (80 registers, 9-11 spills, 7 of them
# kernel-lifetime; cold drain branch) against five (96 registers: base) -- closest at six spills 30-38 and is not a candidate
mkdir -p gpurun_out; OUT=gpurun_out/r06_s6_19.txt; : > $OUT
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base s6 base s6; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c3 --query any --steps 12 --warmup 6
  TAG=$V; Q --config c3 --query first --steps 12 --warmup 6
  TAG=$V; Q --config c3 --query closest --steps 12 --warmup 6
  TAG=$V; Q --config c5s --query any --steps 8 --opt wide=0
  TAG=$V; Q --config c5s --query count --steps 8 --opt wide=0 --rays 20000000
done
cat $OUT
