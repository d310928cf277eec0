#!/bin/bash
# round 6, GPU job 4: the whole parity suite on the tree with the DEEP instantiations at the compiler's budget; A/B against
# round 5's library (TRIRO_ABI_ANY=1: its ABI is 9) and a build with the streaming kernels at five waves per SIMD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_gputest4.txt 2>&1
tail -6 gpurun_out/r06_gputest4.txt
export TRIRO_ABI_ANY=1
timeout 1200 bash scripts/round5/ab.sh gpurun_out/r06_ab4.txt r05 base > gpurun_out/r06_ab4.log 2>&1
AB_SET=stream timeout 600 bash scripts/round5/ab.sh gpurun_out/r06_ab4s.txt base s5 > gpurun_out/r06_ab4s.log 2>&1
cat gpurun_out/r06_ab4.txt gpurun_out/r06_ab4s.txt
