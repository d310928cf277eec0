#!/bin/bash
# round 5: bisect the c5ii emulation's rank-0 step (7.5 ms at the last tree against 1.7 ms two commits earlier) over library variants
OUT=gpurun_out/r05_37; mkdir -p $OUT; : > $OUT/log.txt
REPO=$(pwd)
for V in head 6e2b764 d4b72ee d88dd45 head; do
  if [ $V = head ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  timeout 600 python bench.py --emulate-world 8 --arrival none --records slot --workload c5ii --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=r['emulation']; print('$V', 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], r['verified'])" >> $OUT/log.txt
done
cat $OUT/log.txt
