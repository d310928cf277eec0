#!/bin/bash
# round 6, GPU job 13: the tree with the cold drain branch (direct kernels; streaming closest keeps the plain branch) + the
# native step's side-stream sends and watchdog: the whole GPU suite, bench line, C3 closest / shard check
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r06_gputest13.txt 2>&1
tail -5 gpurun_out/r06_gputest13.txt
python bench.py --no-cpu-baseline --steps 300 > gpurun_out/r06_bench13.json 2> gpurun_out/r06_bench13.err; head -c 400 gpurun_out/r06_bench13.json; echo
AB_SET=stream timeout 600 bash scripts/round5/ab.sh gpurun_out/r06_ab13s.txt base > gpurun_out/r06_ab13s.log 2>&1
cat gpurun_out/r06_ab13s.txt
