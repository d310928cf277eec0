#!/bin/bash
# A/B: the leaf block on every THIRD trip (TR_ALTERNATE=2) instead of every second
REPO=$GRAFT_REPO_ROOT
cd $REPO
TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/alt2/libtriro_hip.so timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -k "steady_state or moving" 2>&1 | tail -2
for V in base alt2 base alt2; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5i --query closest --steps 100 --warmup 40" "--config c5i --query closest --res 512 --steps 100 --warmup 40" "--config c5i --query closest --res 2048 --steps 40 --warmup 30" "--config c2 --query closest --steps 100 --warmup 40" "--config c4 --query closest --steps 100 --warmup 40" "--config room --query closest --steps 100 --warmup 40" "--config c5i --query any --steps 60 --warmup 30" "--config c3 --query any --steps 10 --warmup 4" "--config c3 --query closest --steps 10 --warmup 4" "--config c5s --query closest --steps 10 --warmup 4"; do
    timeout 90 python scripts/run_query.py $A 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
