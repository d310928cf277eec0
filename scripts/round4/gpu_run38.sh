#!/bin/bash
# EIGHT ranks on one GPU over gloo (functional, host-staged): the driver's multi-GPU command lines with the transport replaced
OUT=gpurun_out/r04_run38
mkdir -p $OUT
for args in "" "--scaling strong" "--workload c5ii --total-rays 20000001" "--records packed" "--dst-share 1"; do
  timeout 600 python bench.py --gpus 8 --backend gloo --steps 3 --warmup 1 --min-warmup-ms 0 --no-cpu-baseline --no-companions $args > $OUT/line.json 2> $OUT/err.txt
  echo "rc=$? args=[$args]" >> $OUT/eight_ranks.txt
  python - >> $OUT/eight_ranks.txt <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/r04_run38/line.json").read().strip().splitlines()[-1])
    c = r["config"]
    print("  n_gpus", r["n_gpus"], "scaling", r["scaling"], "verified", r["verified"], "rays_total", c["rays_total"], "shard_rays", c["shard_rays"], "dst_share", c["dst_share"])
    print("  ", c["parallelism"][:260])
except Exception as e:
    print("  no line:", e); print(open("gpurun_out/r04_run38/err.txt").read()[-1500:])
PY
done
cat $OUT/eight_ranks.txt
