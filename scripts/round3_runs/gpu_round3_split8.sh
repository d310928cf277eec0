#!/bin/bash
# A/B: the most expensive sixteenth of the split blocks in EIGHT launch slots (TR_SPLIT8) -- small launches are
# bound by their longest waves
REPO=$GRAFT_REPO_ROOT
cd $REPO
TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/split8/libtriro_hip.so timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -k "steady_state or moving or split or steal" 2>&1 | tail -3
for V in base split8 base split8; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5i --query closest --res 384 --steps 100 --warmup 40" "--config c5i --query closest --res 512 --steps 100 --warmup 40" "--config c5i --query closest --res 640 --steps 100 --warmup 40" "--config c5i --query closest --res 768 --steps 100 --warmup 40" "--config c5i --query closest --steps 100 --warmup 40" "--config c4 --query closest --res 512 --steps 100 --warmup 40" "--config c4 --query closest --steps 100 --warmup 40" "--config c2 --query closest --res 512 --steps 100 --warmup 40" "--config c2 --query closest --steps 100 --warmup 40" "--config room --query closest --steps 100 --warmup 40" "--config c5i --query any --res 512 --steps 60 --warmup 30" "--config c5i --query count --res 512 --steps 40 --warmup 20"; do
    timeout 90 python scripts/run_query.py $A 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done
