#!/bin/bash
# round 5: the deferred sort as a workgroup of the next launch (sort_inline): parity, then A/B on the headline and the image configs
# against the library of the previous commit (lib_var/r05m) and against sort_inline=0 of this one
OUT=gpurun_out/r05_16
mkdir -p $OUT; rm -f $OUT/ab.txt
REPO=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -5 $OUT/pytest.txt
for rep in 1 2 3; do
for V in old 1 0; do
  if [ $V = old ]; then export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/r05m/libtriro_hip.so; O=""; else unset TRIRO_HIP_LIBRARY; O="--opt sort_inline=$V"; fi
  timeout 300 python bench.py --steps 400 --warmup 50 --no-companions --no-cpu-baseline $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V headline', r['value'], r['ms_per_step'], r['roofline']['kernel_avg_ms'], r['verified'])" >> $OUT/ab.txt
done
done
for V in old 1 0; do
  if [ $V = old ]; then export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/r05m/libtriro_hip.so; O=""; else unset TRIRO_HIP_LIBRARY; O="--opt sort_inline=$V"; fi
  for A in "--config c2 --query closest" "--config c4 --query closest" "--config c5i --query first" "--config c5i --query any" "--config terrain --query closest" "--config room --query closest" "--config c5i --res 512 --query closest"; do
    python scripts/run_query.py $A --steps 200 --warmup 40 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
