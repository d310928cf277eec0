#!/bin/bash
mkdir -p gpurun_out
REPO=$GRAFT_REPO_ROOT
for V in base se1 se7 si4 si8 sh0 base; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5i --query closest" "--config c5i --res 512 --query closest" "--config c2 --query closest" "--config c4 --query closest" "--config room --query closest" "--config c5i --query any"; do
    [ "$A" = "--config c5i --rays-hash" ] && continue
    timeout 90 python scripts/run_query.py --steps 60 --warmup 24 $A 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])"
  done
done | tee gpurun_out/r3s_steal_variants.txt
