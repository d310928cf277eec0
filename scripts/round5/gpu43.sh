#!/bin/bash
# round 5: the query- and mesh-aware streaming boundary (tr_policy::stream_min_rays): full GPU suite, then the automatic policy at the sizes around the old and new boundaries
OUT=gpurun_out/r05_43; mkdir -p $OUT; : > $OUT/auto.txt
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.txt
for C in c3 c5s; do for Q in closest any count; do for N in 1500000 2200000 3000000 4194304 10000000; do
  python scripts/run_query.py --config $C --query $Q --rays $N --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$C', r['rays'], '$Q', 'auto', r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/auto.txt
done; done; done
cat $OUT/auto.txt
