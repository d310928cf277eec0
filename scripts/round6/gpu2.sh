#!/bin/bash
# round 6, GPU job 2: contract 3 with the float64 part parked and decided at the end of a trip (tr_drain_exact), staged early
# exits in the float32 part, six waves on the direct kernels -- parity suite + A/B against round 5's library
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r06_gputest2.txt 2>&1
tail -5 gpurun_out/r06_gputest2.txt
timeout 1500 bash scripts/round5/ab.sh gpurun_out/r06_ab2.txt r05 base > gpurun_out/r06_ab2.log 2>&1
cat gpurun_out/r06_ab2.txt
