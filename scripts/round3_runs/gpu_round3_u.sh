#!/bin/bash
# the N > 1 result pipeline on ONE GPU (self-gather): what packed trace + exchange + expansion cost next to the plain call
B="python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 1 --no-cpu-baseline --no-companions"
P='import sys,json
for ln in sys.stdin:
    if ln.startswith("{\"metric\""):
        r=json.loads(ln); print(TAG, r["value"], "Mrays/s", r["ms_per_step"], "ms/step", r["verified"], r["config"]["parallelism"][:60])'
for A in "--steps 300" "--steps 300 --force-gather" "--workload c5ii --steps 10 --warmup 2" "--workload c5ii --steps 10 --warmup 2 --force-gather" "--workload c5ii --steps 10 --warmup 2 --force-gather --chunks 8" "--workload c5ii --steps 10 --warmup 2 --force-gather --chunks 1"; do
  timeout 400 $B $A 2>/dev/null | python3 -c "TAG='''$A'''
$P"
done
