#!/bin/bash
# round 5: from which trip should a wave of unrelated rays give subtrees away?  TR_STEAL_EARLY = 8 / 12 / 16 (shipped) / 24, incoherent batches, direct launch
OUT=gpurun_out/r05_56; mkdir -p $OUT; : > $OUT/ab.txt
REPO=$(pwd)
for rep in 1 2; do for V in base se8 se4; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for C in c3 c5s; do for Q in closest any; do for N in 1048576 2200000; do
    python scripts/run_query.py --config $C --query $Q --rays $N --steps 40 --warmup 15 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', '$C', r['rays'], '$Q', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
  done; done; done
done; done
sort -k2,4 -s $OUT/ab.txt
