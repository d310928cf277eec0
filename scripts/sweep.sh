#!/bin/bash
# On the GPU box: run bench.py for several option sets; one line each.  usage: scripts/sweep.sh tag "args1" "args2" ...
TAG=$1; shift
mkdir -p gpurun_out
: > gpurun_out/sweep_$TAG.log
for A in "$@"; do
  OUT=$(timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $A 2>&1 | tail -1)
  echo "$A => $(echo "$OUT" | python3 -c "
import sys,json
try:
    r=json.loads(sys.stdin.read()); print(r['value'],'Mrays/s kernel_avg_ms',r['roofline']['kernel_avg_ms'],'min',r['roofline']['kernel_min_ms'], r.get('trace_stats',''))
except Exception as e: print('FAIL',e)
")" | tee -a gpurun_out/sweep_$TAG.log
done
