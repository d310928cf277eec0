#!/bin/bash
# round 5: waves of unrelated rays steal from trip 16 on (the wave decides): parity, the image configs (must not move), incoherent batches, then the
# streaming boundary again (the direct launch it is measured against got faster)
OUT=gpurun_out/r05_52; mkdir -p $OUT; : > $OUT/ab.txt
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_configs.py tests/test_gpu_round2.py tests/test_gpu_interior.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
python bench.py --steps 400 --warmup 50 --no-companions --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', r['value'], r['ms_per_step'], r['verified'])" >> $OUT/ab.txt
for A in "--config c2 --query closest" "--config c4 --query closest" "--config c5i --query any" "--config c5i --query first" "--config terrain --query closest" "--config room --query closest" "--config c5i --res 512 --query closest" "--config c5i --query closest --flat"; do
  python scripts/run_query.py $A --steps 100 --warmup 40 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('image', r['config'], r['query'], r['rays'], r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
done
for C in c3 c5s; do for Q in closest any first; do for N in 262144 1048576 1500000 2200000 3000000 4194304 6000000; do for O in "--opt stream=0" "--opt stream=2"; do
  [ $N -le 1048576 ] && [ "$O" = "--opt stream=2" ] && continue
  python scripts/run_query.py --config $C --query $Q --rays $N --steps 24 --warmup 10 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$C', r['rays'], '$Q', '$O', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
done; done; done; done
cat $OUT/ab.txt
