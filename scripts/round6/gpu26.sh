#!/bin/bash
# round 6, GPU job 26: arrays above 4 GiB (84 M triangles, TRIRO_TEST_HUGE=1: the instantiations with 64-bit addressing, back at
# the compiler's register budget) + the rest of the large-mesh tests; 3 more fuzz seeds
mkdir -p gpurun_out
TRIRO_TEST_HUGE=1 timeout 1500 python -m pytest tests/test_gpu_round2.py -q -p no:cacheprovider -k "large_meshes" > gpurun_out/r06_gputest_huge.txt 2>&1; tail -3 gpurun_out/r06_gputest_huge.txt
for S in 611 612 613; do
  timeout 1200 python scripts/fuzz_parity.py --iters 200 --seed $S > gpurun_out/r06_fuzz_seed$S.txt 2>&1; tail -1 gpurun_out/r06_fuzz_seed$S.txt
done
