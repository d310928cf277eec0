#!/bin/bash
# round 5: the coherence probe's verdict as a hint (probe_hint): parity, then A/B on the incoherent configs
OUT=gpurun_out/r05_26
mkdir -p $OUT; rm -f $OUT/ab.txt
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_configs.py tests/test_gpu_wide.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -4 $OUT/pytest.txt
for rep in 1 2; do
for V in 1 0; do
  for A in "--config c3 --query any" "--config c3 --query closest" "--config c5s --query closest" "--config c5s --query count" "--config c5s --query any" "--config c5i --res 2048 --query closest"; do
    python scripts/run_query.py $A --steps 40 --warmup 10 --opt probe_hint=$V 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('probe_hint=$V', r['config'], r['query'], r['rays'], r['tris'], r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab.txt
  done
done
done
sort -k2,3 -s $OUT/ab.txt
