#!/bin/bash
OUT=gpurun_out/r06_final7
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -4 $OUT/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt | cut -c1-100
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 300 $OUT/bench_default.json; echo
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench_default.err; head -c 300 $OUT/bench_driver_args.json; echo
