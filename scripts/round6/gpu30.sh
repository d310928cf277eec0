#!/bin/bash
# round 6, GPU job 30: the stealing count launch at SEVEN waves per SIMD again (round 5: -3 %, not shipped because the DEEP
# instantiation faulted) -- now only its 32-bit instantiations (72 registers, 8-10 kernel-lifetime spills, none in the
# trips), the 64-bit ones at the compiler's budget: lib_var/c7 against the shipped library
mkdir -p gpurun_out; OUT=gpurun_out/r06_c7_30.txt; : > $OUT
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base c7 base c7; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c4 --query count --steps 30 --warmup 20
  TAG=$V; Q --config c5i --query count --steps 30 --warmup 20
  TAG=$V; Q --config c2 --query count --steps 30 --warmup 20
  TAG=$V; Q --config terrain --query count --steps 30 --warmup 20
  TAG=$V; Q --config room --query count --steps 30 --warmup 20
  TAG=$V; Q --config c5i --res 2048 --query count --steps 10 --warmup 6
  TAG=$V; Q --config c3 --query count --rays 2000000 --steps 10 --warmup 6
done
cat $OUT
export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/c7/libtriro_hip.so
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "count or contains or soup or terrain or interior or c4 or usteal or robust" 2>&1 | tail -3
