#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider -k "rccl_one_rank" > gpurun_out/r06_gputest28.txt 2>&1; tail -25 gpurun_out/r06_gputest28.txt | cut -c1-400
