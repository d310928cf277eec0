#!/bin/bash
# round 5: the direct launch on the 8-wide nodes after the fused decode -- is count (and closest) better off there now?  + suite on the current tree
OUT=gpurun_out/r05_12
mkdir -p $OUT
for C in c4 c5i terrain c2 room; do for Q in count location closest; do for WD in 0 1 2 3; do
  python scripts/run_query.py --config $C --query $Q --steps 40 --warmup 20 --opt wide_direct=$WD 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])" >> $OUT/wide_direct.txt
done; done; done
cat $OUT/wide_direct.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
