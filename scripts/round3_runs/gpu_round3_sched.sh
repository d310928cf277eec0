#!/bin/bash
# A/B: LLVM machine-scheduler strategies / -O2 for the whole library
REPO=$GRAFT_REPO_ROOT
cd $REPO
for V in base maxilp maxmem iterilp o2 base maxilp maxmem iterilp o2; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5i --query closest --steps 100 --warmup 40" "--config c5i --query closest --res 512 --steps 100 --warmup 40" "--config c2 --query closest --steps 100 --warmup 40" "--config c4 --query closest --steps 100 --warmup 40" "--config c4 --query count --steps 30 --warmup 12" "--config room --query closest --steps 100 --warmup 40" "--config c3 --query any --steps 10 --warmup 4" "--config c5s --query closest --steps 10 --warmup 4"; do
    timeout 90 python scripts/run_query.py $A 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
