#!/bin/bash
# round 5: block costs recorded cost of a split block: x 2^(lg parts + 1) (base) against x 2^(lg parts) (st0): steady state against a moving camera
OUT=gpurun_out/r05_24
mkdir -p $OUT; rm -f $OUT/ab.txt
REPO=$(pwd)
for rep in 1 2 3; do
for V in base st0; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V headline', r['value'], r['ms_per_step'], 'moving', r['roofline']['moving_camera_kernel_ms'], 'cold', r['roofline']['cold_kernel_ms'], r['verified'])" >> $OUT/ab.txt
done
done
for V in base st0; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c2 --query closest" "--config c4 --query closest" "--config c5i --query any" "--config terrain --query closest" "--config room --query closest" "--config c4 --query count"; do
    python scripts/run_query.py $A --steps 200 --warmup 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
