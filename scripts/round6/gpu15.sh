#!/bin/bash
# round 6, GPU job 15: VERDICT r05 "next" #7 again -- the DEEP instantiations (64-bit trail words, hierarchies of more than
# 32 levels) held to six waves per SIMD NOW THAT the spills sit around the float64 call only (cold drain branch): does the
# 21 M-triangle mesh still fault?  And what does the sixth wave buy on a 5.2 M-triangle mesh (34 levels)?
mkdir -p gpurun_out
OUT=gpurun_out/r06_deep15.txt; : > $OUT
export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/deep6/libtriro_hip.so
echo "# deep6: tests/test_gpu_round2.py -k large_meshes" >> $OUT
timeout 900 python -m pytest tests/test_gpu_round2.py -q -p no:cacheprovider -k large_meshes >> $OUT 2>&1; echo "rc=$?" >> $OUT
python -c "import torch; x=torch.ones(4,device='cuda'); print('gpu alive', float(x.sum()))" >> $OUT 2>&1
unset TRIRO_HIP_LIBRARY
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
for V in base deep6 base deep6; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c5i --subdiv 9 --query closest --steps 40 --warmup 30
  TAG=$V; Q --config c5i --subdiv 9 --query first --steps 40 --warmup 30
  TAG=$V; Q --config c5i --subdiv 9 --query any --steps 40 --warmup 30
done
cat $OUT | grep -v amdgpu.ids | tail -30
