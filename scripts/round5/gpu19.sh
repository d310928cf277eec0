#!/bin/bash
# round 5: the unordered count launch (78 registers = 6 waves per SIMD since the fused box test) forced to 7 waves (c7: 72
# registers, 5 spills outside the trips) and, with a 4-entry leaf queue (10 KiB of LDS per workgroup), to 8 (c8: 64 registers, 19 spills)
OUT=gpurun_out/r05_19
mkdir -p $OUT; rm -f $OUT/ab.txt
REPO=$(pwd)
for rep in 1 2; do
for V in base c7 c8; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c4 --query count" "--config c5i --query count" "--config terrain --query count" "--config room --query count" "--config c2 --query count" "--config c2 --query location" "--config c4 --query count --res 512" "--config c5s --query count --steps 8 --warmup 4 --opt stream=0"; do
    python scripts/run_query.py --steps 60 --warmup 30 $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab.txt
  done
done
done
sort -k2,3 -s $OUT/ab.txt
