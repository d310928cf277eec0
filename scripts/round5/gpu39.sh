#!/bin/bash
# round 5: the stealing count launch carries the deferred sort too: parity, A/B
OUT=gpurun_out/r05_39; mkdir -p $OUT; rm -f $OUT/ab.txt
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_interior.py tests/test_gpu_terrain.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
for rep in 1 2; do for V in 1 0; do
  for A in "--config c4 --query count" "--config c5i --query count" "--config terrain --query count" "--config room --query count" "--config c2 --query count"; do
    python scripts/run_query.py $A --steps 200 --warmup 40 --opt sort_inline=$V 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('sort_inline=$V', r['config'], r['query'], r['rays'], r['tris'], r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab.txt
  done
done; done
sort -k2,3 -s $OUT/ab.txt
