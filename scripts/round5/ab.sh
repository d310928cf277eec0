#!/bin/bash
# A/B of library variants built into trimesh-ray-optix_amd/lib_var/<name>/ over the configs of DESIGN.md section 5
# usage: scripts/round5/ab.sh OUTFILE base vA vB ...     (one text line per (variant, config, query))
REPO=$(pwd)
OUT=$1; shift
for V in "$@"; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for k in 1 2; do
  python bench.py --no-cpu-baseline --no-companions --steps 300 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V headline', r['value'], r['roofline']['kernel_avg_ms'])" >> $OUT
  done
  if [ "${AB_SET:-all}" = stream ]; then
  for A in "--config c3 --query any --steps 12 --warmup 6" "--config c3 --query closest --steps 12 --warmup 6" "--config c5s --query closest --steps 12 --warmup 6" \
           "--config c5s --query count --steps 8" "--config c5s --query closest --steps 8 --subdiv 9" "--config c5i --res 4096 --query closest --steps 8"; do
  python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT
  done
  continue
  fi
  if [ "${AB_SET:-all}" = direct ]; then
  for A in "--config c2 --query closest --steps 60 --warmup 40" "--config c4 --query closest --steps 60 --warmup 40" "--config c5i --query any --steps 60 --warmup 40" \
           "--config c5i --query first --steps 60 --warmup 40" "--config terrain --query closest --steps 60 --warmup 40" "--config room --query closest --steps 60 --warmup 40" \
           "--config c5i --res 2048 --query closest --steps 20 --warmup 20" "--config c5i --res 512 --query closest --steps 60 --warmup 40"; do
  python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT
  done
  continue
  fi
  for A in "--config c2 --query closest" "--config c4 --query closest" "--config c4 --query count" "--config c4 --query location" \
           "--config c5i --query any" "--config c5i --query count" "--config c5i --query location" \
           "--config c5i --res 2048 --query closest --steps 8" "--config c5i --res 4096 --query closest --steps 8" \
           "--config c3 --query any --steps 8" "--config c3 --query closest --steps 8" "--config c5s --query closest --steps 8" \
           "--config c5s --query count --steps 8" "--config c5s --query closest --steps 8 --opt wide=1" "--config c3 --query any --steps 8 --opt wide=1" \
           "--config c5s --query closest --steps 8 --subdiv 9" "--config terrain --query closest" "--config terrain --query location" \
           "--config room --query closest" "--config soup --query location --steps 8"; do
  python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT
  done
done
