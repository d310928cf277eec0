#!/bin/bash
# round 5: static tail split -- parity first, then what it buys the unlearned launch; grid nodes as the starting flavour
OUT=gpurun_out/r05_6
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_r5.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r5.txt; tail -5 $OUT/pytest_r5.txt
timeout 500 python scripts/fuzz_parity.py --iters 100 --seed 502 > $OUT/fuzz_502.txt 2>&1; tail -1 $OUT/fuzz_502.txt
Q="python scripts/run_query.py --query closest --steps 40 --warmup 10"
for C in "c5i" "c4" "c2" "terrain" "room"; do
for O in "--opt adaptive=0 --opt tail_split=0" "--opt adaptive=0 --opt tail_split=20" "--opt adaptive=0 --opt tail_split=35" "--opt adaptive=0 --opt tail_split=50" "--opt adaptive=0 --opt tail_split=70" \
         "--opt adaptive=0 --opt grid_nodes=2 --opt tail_split=0" "--opt adaptive=0 --opt grid_nodes=2 --opt tail_split=35" "--opt adaptive=0 --opt grid_nodes=2 --opt tail_split=50" "--opt adaptive=0 --opt grid_nodes=2 --opt tail_split=70"; do
  $Q --config $C $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])" >> $OUT/cold_tail.txt
done; done
cat $OUT/cold_tail.txt
# node flavour: exact (0) vs grid (2) vs measured (1), steady state
for C in "c5i" "c4" "c2" "terrain" "room" "soup"; do for G in 0 1 2; do
  python scripts/run_query.py --config $C --query closest --steps 60 --warmup 60 --opt grid_nodes=$G 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])" >> $OUT/flavour.txt
done; done
python scripts/run_query.py --config c5i --query first --steps 60 --warmup 60 --opt grid_nodes=0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])" >> $OUT/flavour.txt
python scripts/run_query.py --config c5i --query first --steps 60 --warmup 60 --opt grid_nodes=2 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'])" >> $OUT/flavour.txt
cat $OUT/flavour.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print({k:r[k] for k in ('value','value_warmup_requested')}, {k:v for k,v in r['roofline'].items() if 'ms' in k})"
