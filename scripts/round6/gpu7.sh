#!/bin/bash
# round 6, GPU job 7: the inside test as ONE decision (no exits after U / after V) -- A/B against round 5's library and a build
# with the culling limit kept in a register (limh); SQ counters of the headline launch; a quick parity check
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round6.py -m gpu -q -p no:cacheprovider -x -k "not 2_to_the_32" > gpurun_out/r06_gputest7.txt 2>&1
tail -3 gpurun_out/r06_gputest7.txt
export TRIRO_ABI_ANY=1
timeout 1500 bash scripts/round5/ab.sh gpurun_out/r06_ab7.txt r05 base limh > gpurun_out/r06_ab7.log 2>&1
bash scripts/pmc_quick.sh r06_base7 "--no-companions" > gpurun_out/r06_pmc_base7.txt 2>&1
cat gpurun_out/r06_ab7.txt gpurun_out/r06_pmc_base7.txt
