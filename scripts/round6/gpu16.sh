#!/bin/bash
# round 6, GPU job 16: the tree with the DEEP instantiations at six waves (spills around the float64 call only): the whole
# GPU suite, 2 x 200 fuzz iterations (deep chain meshes among the families), DEEP timings, bench line
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_gputest16.txt 2>&1
tail -5 gpurun_out/r06_gputest16.txt
for S in 601 602; do
  timeout 1500 python scripts/fuzz_parity.py --iters 200 --seed $S > gpurun_out/r06_fuzz_seed$S.txt 2>&1; tail -2 gpurun_out/r06_fuzz_seed$S.txt
done
python bench.py --no-cpu-baseline --steps 300 > gpurun_out/r06_bench16.json 2> gpurun_out/r06_bench16.err; head -c 300 gpurun_out/r06_bench16.json; echo
