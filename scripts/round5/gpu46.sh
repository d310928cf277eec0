#!/bin/bash
# round 5: the automatic policy after "8-wide nodes only from 8 M rays on" + "streaming from 2 M rays on large meshes again": incoherent batches, three meshes
OUT=gpurun_out/r05_47; mkdir -p $OUT; : > $OUT/auto.txt
for SD in 8 9; do for Q in closest any first; do for N in 2200000 3000000 4194304 6000000 12500000; do
  python scripts/run_query.py --config c5s --subdiv $SD --query $Q --rays $N --steps 14 --warmup 6 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5s', r['tris'], r['rays'], '$Q', 'auto', r['ms_mean'], r['ms_min'])" >> $OUT/auto.txt
done; done; done
for N in 6000000 12500000; do python scripts/run_query.py --config c5s --query count --rays $N --steps 12 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5s', r['tris'], r['rays'], 'count', 'auto', r['ms_mean'], r['ms_min'])" >> $OUT/auto.txt; done
cat $OUT/auto.txt
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.txt
