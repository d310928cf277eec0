#!/bin/bash
# round 6, GPU job 8: the native N-rank step (functional test on one GPU incl. RCCL loopback, host time per step); what the
# float64 part of the predicate costs (noexact: undecided tests count as misses -- timing only)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -p no:cacheprovider -x -k "native_step" > gpurun_out/r06_gputest8.txt 2>&1
tail -15 gpurun_out/r06_gputest8.txt
timeout 600 python scripts/profile_pipeline_host.py > gpurun_out/r06_pipeline_host.txt 2>&1
tail -4 gpurun_out/r06_pipeline_host.txt
export TRIRO_ABI_ANY=1
AB_SET=direct timeout 900 bash scripts/round5/ab.sh gpurun_out/r06_ab8.txt r05 base noexact > gpurun_out/r06_ab8.log 2>&1
cat gpurun_out/r06_ab8.txt
