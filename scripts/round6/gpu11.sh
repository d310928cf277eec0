#!/bin/bash
# round 6, GPU job 11: compensated ray anchoring + the live frame: parity suite, A/B against round 5, far-camera series
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_gputest11.txt 2>&1
tail -8 gpurun_out/r06_gputest11.txt
export TRIRO_ABI_ANY=1
timeout 1200 bash scripts/round5/ab.sh gpurun_out/r06_ab11.txt r05 base > gpurun_out/r06_ab11.log 2>&1
OUT=gpurun_out/r06_far11.txt; : > $OUT
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for far in 1 10 100 1000 10000; do
  unset TRIRO_HIP_LIBRARY; TAG="base far=$far"; Q --config c5i --query closest --steps 40 --warmup 30 --far $far
  TAG="base far=$far"; Q --config c5i --query count --steps 20 --warmup 10 --far $far
  export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/r05/libtriro_hip.so; TAG="r05 far=$far"; Q --config c5i --query closest --steps 40 --warmup 30 --far $far
  TAG="r05 far=$far"; Q --config c5i --query count --steps 20 --warmup 10 --far $far
done
cat gpurun_out/r06_ab11.txt $OUT
