#!/bin/bash
# On the GPU box: GPU parity tests then the headline bench.  usage: scripts/gpu_check.sh <tag> [bench args]
TAG=${1:-x}; shift || true
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_$TAG.log
timeout 600 python bench.py --steps 30 --warmup 5 --stats --no-cpu-baseline $* > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_$TAG.log | python3 -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); print('VALUE',r['value'],r['unit'],'ms/step',r['ms_per_step'],'kernel_avg_ms',r['roofline']['kernel_avg_ms'],'min',r['roofline']['kernel_min_ms'],r.get('trace_stats'))
except Exception as e: print('parse fail',e)
"
