#!/bin/bash
# round 5, third GPU run: suite on the tree with the six-wave streaming launch and the sched_acquire fix; headline profile; wide occupancy A/B
OUT=gpurun_out/r05_3
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
bash scripts/profile_bench.sh r05a --steps 1000 --warmup 50 --no-companions > $OUT/profile_bench.txt 2>&1
PROFILE_STEPS=1000 python scripts/summarize_profile.py r05a > $OUT/r05a_summary.txt 2>&1; tail -30 $OUT/r05a_summary.txt
REPO=$(pwd)
for V in base ww5; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5s --query closest --steps 12 --warmup 6 --opt wide=1" "--config c5s --query closest --steps 12 --warmup 6 --subdiv 9" \
           "--config c5s --query count --steps 12 --warmup 6" "--config c3 --query any --steps 12 --warmup 6 --opt wide=1" \
           "--config c5s --query closest --steps 12 --warmup 6" "--config c3 --query any --steps 12 --warmup 6"; do
    python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab_wide_waves.txt
  done
done
cat $OUT/ab_wide_waves.txt
