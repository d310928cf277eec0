#!/bin/bash
# A/B: launches of <= 2048 blocks may split HALF of their blocks (shift 1) instead of a quarter
REPO=$GRAFT_REPO_ROOT
cd $REPO
for V in base shift1 base shift1; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5i --query closest --res 384 --steps 100 --warmup 40" "--config c5i --query closest --res 512 --steps 100 --warmup 40" "--config terrain --query closest --res 640 --steps 100 --warmup 40" "--config c4 --query closest --res 512 --steps 100 --warmup 40" "--config c2 --query closest --res 512 --steps 100 --warmup 40" "--config room --query closest --steps 100 --warmup 40" "--config c5i --query any --res 512 --steps 60 --warmup 30"; do
    timeout 90 python scripts/run_query.py $A 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
