#!/bin/bash
# round 4, GPU call 8: the whole GPU suite, the bench line + its rocprof summary, policy matrices, two-halves experiment
OUT=gpurun_out/r04_run8
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -6 $OUT/pytest_gpu_full.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 400 $OUT/bench_default.json; echo
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench_default.err; head -c 300 $OUT/bench_driver_args.json; echo
timeout 300 python scripts/round4/exp_two_halves.py > $OUT/two_halves.json 2> $OUT/two_halves.err; cat $OUT/two_halves.json
bash scripts/profile_bench.sh r04m --steps 1000 --warmup 50 --no-companions > $OUT/profile_bench.txt 2>&1
bash scripts/policy_matrix.sh terrain "640 1024 1920" > $OUT/policy_terrain.txt 2>&1
bash scripts/policy_matrix.sh soup "640 1024" > $OUT/policy_soup.txt 2>&1
bash scripts/policy_matrix.sh c2 "512 1024" > $OUT/policy_c2.txt 2>&1
tail -8 $OUT/policy_terrain.txt
