#!/bin/bash
# end-of-round validation: the whole GPU suite, smoke, the bench line (default and the driver's arguments), the rocprof summary
OUT=gpurun_out/r04_final
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -5 $OUT/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 260 $OUT/bench_default.json; echo
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench_default.err; head -c 200 $OUT/bench_driver_args.json; echo
bash scripts/profile_bench.sh r04m --steps 1000 --warmup 50 --no-companions > $OUT/profile_bench.txt 2>&1
