#!/bin/bash
# round 5: re-tune what the cheaper box test may have shifted: leaf-block cadence, look cadence (compile-time), steal thresholds, refill thresholds (options)
OUT=gpurun_out/r05_10
mkdir -p $OUT
REPO=$(pwd)
for V in base alt2 alt0 se1 se7; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5i --query closest --steps 100 --warmup 60" "--config c4 --query closest --steps 60 --warmup 40" "--config c2 --query closest --steps 60 --warmup 40" \
           "--config c5s --query closest --steps 12 --warmup 6 --opt wide=0" "--config c3 --query any --steps 12 --warmup 6"; do
    python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])" >> $OUT/variants.txt
  done
done
cat $OUT/variants.txt
unset TRIRO_HIP_LIBRARY
for O in "" "--opt steal=48" "--opt steal=80" "--opt steal=112" "--opt split_steal=4" "--opt split_steal=16" "--opt split_floor=20" "--opt split_floor=80" "--opt xcd_chunk=64" "--opt xcd_chunk=256"; do
  for C in c5i c4; do
    python scripts/run_query.py --config $C --query closest --steps 100 --warmup 60 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])" >> $OUT/options.txt
  done
done
for O in "" "--opt stream_refill=24" "--opt stream_refill=40" "--opt stream_refill=48" "--opt stream_rays=128" "--opt stream_rays=512"; do
  for C in "c5s --query closest" "c3 --query any"; do
    python scripts/run_query.py --config $C --steps 12 --warmup 6 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])" >> $OUT/options.txt
  done
done
cat $OUT/options.txt
