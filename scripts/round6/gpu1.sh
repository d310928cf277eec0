#!/bin/bash
# round 6, first GPU job: contract 3 (watertight inside test, no own-box clamp) -- parity suite + A/B against round 5's library
# variants: r05 = round 5's tree, base = contract 3 with the compiler's register budget, w6 = ... with six waves forced on the direct kernels
mkdir -p gpurun_out
timeout 1500 bash scripts/round5/ab.sh gpurun_out/r06_ab1.txt r05 base w6 > gpurun_out/r06_ab1.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_gputest1.txt 2>&1
tail -5 gpurun_out/r06_gputest1.txt
cat gpurun_out/r06_ab1.txt
