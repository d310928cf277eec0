#!/bin/bash
# round 6, GPU job 27: count launches list their undecided leaf tests and k_count_resolve decides them (count_defer = 1: base)
# against deciding them in the launch (count_defer = 0, and the library of the previous commit: r06prev); parity first
mkdir -p gpurun_out; OUT=gpurun_out/r06_defer27.txt; : > $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "count or contains or soup or terrain or interior or c4 or fuzz or round3 or usteal or graph or robust" > gpurun_out/r06_gputest27.txt 2>&1; tail -3 gpurun_out/r06_gputest27.txt
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base nodefer r06prev base nodefer r06prev; do
  X=""; unset TRIRO_HIP_LIBRARY
  if [ $V = nodefer ]; then X="--opt count_defer=0"; fi
  if [ $V = r06prev ]; then export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/r06prev/libtriro_hip.so; fi
  TAG=$V; Q --config c4 --query count --steps 30 --warmup 20 $X
  TAG=$V; Q --config c5i --query count --steps 30 --warmup 20 $X
  TAG=$V; Q --config c2 --query count --steps 30 --warmup 20 $X
  TAG=$V; Q --config terrain --query count --steps 30 --warmup 20 $X
  TAG=$V; Q --config room --query count --steps 30 --warmup 20 $X
  TAG=$V; Q --config c5i --res 2048 --query count --steps 10 --warmup 6 $X
  TAG=$V; Q --config c3 --query count --rays 2000000 --steps 10 --warmup 6 $X
done
cat $OUT
