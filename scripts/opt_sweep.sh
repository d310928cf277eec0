#!/bin/bash
# usage: scripts/opt_sweep.sh name v1 v2 ... ; runs the headline, 2048^2 and hash workloads per value
name=$1; shift
for v in "$@"; do
  for a in "" "--res 2048" "--rays hash" "--res 512" "--res 4096"; do
    python bench.py --no-cpu-baseline --opt $name=$v $a 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$name=$v','$a',j['value'],j['roofline']['kernel_avg_ms'])"
  done
done
