#!/bin/bash
# A/B of library variants built into trimesh-ray-optix_amd/lib_var/<name>/ (usage: scripts/ab_variant.sh base vA vB ...)
REPO=$(pwd)
for V in "$@"; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for k in 1 2; do
  python bench.py --no-cpu-baseline --no-companions --steps 300 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V headline', r['value'], r['roofline']['kernel_avg_ms'])"
  done
  for A in "--config c2 --query closest" "--config c4 --query closest" "--config c4 --query count" "--config c5i --query any" "--config c5i --res 2048 --query closest --steps 8" "--config c5i --res 4096 --query closest --steps 8" "--config c3 --query any --steps 8" "--config c3 --query closest --steps 8" "--config c5s --query closest --steps 8"; do
  python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['ms_mean'], r['mrays_per_s'])"
  done
done
