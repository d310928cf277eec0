#!/bin/bash
# round 5: USTEAL_FROM_TRIP = 4: the count tests, then count direct (stream=0) against streamed (stream=2) at the sizes around the 8 M boundary
OUT=gpurun_out/r05_59; mkdir -p $OUT; : > $OUT/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.txt
for C in c3 c5s; do for N in 6000000 10000000 12500000 16777216 25000000; do for O in "--opt stream=0" "--opt stream=2"; do
  python scripts/run_query.py --config $C --query count --rays $N --steps 10 --warmup 4 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$C', r['rays'], 'count', '$O', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
done; done; done
cat $OUT/ab.txt
