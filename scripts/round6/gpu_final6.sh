#!/bin/bash
# contract 3 with ONE OWNER PER EDGE on exact ties: GPU suite, bench line, A/B of the direct configs against the previous library, fuzz
OUT=gpurun_out/r06_final6
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -6 $OUT/pytest_gpu_full.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 300 $OUT/bench_default.json; echo
AB_SET=direct timeout 900 bash scripts/round5/ab.sh $OUT/ab.txt prev base > $OUT/ab.log 2>&1
Q() { python scripts/run_query.py "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$TAG', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab.txt; }
for V in prev base; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  TAG=$V; Q --config c4 --query count --steps 30 --warmup 20
  TAG=$V; Q --config c5i --query count --steps 30 --warmup 20
  TAG=$V; Q --config c4 --query location --steps 20 --warmup 10
  TAG=$V; Q --config c3 --query any --steps 12 --warmup 6
  TAG=$V; Q --config c3 --query closest --steps 12 --warmup 6
  TAG=$V; Q --config c5s --query closest --steps 8
done
unset TRIRO_HIP_LIBRARY
cat $OUT/ab.txt
for S in 641 642; do timeout 900 python scripts/fuzz_parity.py --iters 200 --seed $S > $OUT/fuzz_seed$S.txt 2>&1; tail -1 $OUT/fuzz_seed$S.txt; done
