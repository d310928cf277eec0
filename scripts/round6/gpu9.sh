#!/bin/bash
# round 6, GPU job 9: (1) rank 0's share of the weak-scaling emulation after the contract change (expansion got slower: float64
# barycentrics); (2) the stealing launch against the plain one at 16.7 M rays (the plain launch lost its cheap leaf test);
# (3) the same image from 1x / 10x / 100x / 1000x the camera distance (the inside test's margin grows with it), round 5 beside
mkdir -p gpurun_out; OUT=gpurun_out/r06_job9.txt; : > $OUT
for s in auto 0.30 0.22 0.15; do
  timeout 600 python bench.py --emulate-world 8 --workload c5i --steps 60 --warmup 20 --dst-share $s 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); e=r['emulation']; print('c5i weak share $s', r['config']['dst_share'], 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], 'plain', e['plain_1gpu_ms_per_step'], 'implied', e['implied_scaling_vs_1gpu'], r['verified'])" >> $OUT
done
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT; }
TAG=base; Q --config c5i --res 4096 --query closest --steps 8
TAG=base; Q --config c5i --res 4096 --query closest --steps 8 --opt steal=64
TAG=base; Q --config c5i --res 2896 --query closest --steps 8
TAG=base; Q --config c5i --res 2896 --query closest --steps 8 --opt steal=64
export TRIRO_ABI_ANY=1
for far in 1 10 100 1000 10000; do
  unset TRIRO_HIP_LIBRARY; TAG="base far=$far"; Q --config c5i --query closest --steps 40 --warmup 30 --far $far
  TAG="base far=$far"; Q --config c5i --query count --steps 20 --warmup 10 --far $far
  export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/r05/libtriro_hip.so; TAG="r05 far=$far"; Q --config c5i --query closest --steps 40 --warmup 30 --far $far
  TAG="r05 far=$far"; Q --config c5i --query count --steps 20 --warmup 10 --far $far
done
cat $OUT
