#!/bin/bash
OUT=gpurun_out/r04_run36
mkdir -p $OUT
for opts in "" "--opt stream=2" "--opt stream=2 --opt stream_rays=128" "--opt stream=2 --opt stream_rays=64" "--opt stream=2 --opt stream_rays=64 --opt stream_refill=16" "--opt stream=2 --opt stream_rays=128 --opt stream_refill=16" "--opt stream=2 --opt stream_rays=128 --opt stream_dynamic=0" "--opt stream=2 --opt stream_rays=64 --opt stream_dynamic=0 --opt stream_refill=16"; do
  python scripts/run_query.py --config c5i --query closest --steps 20 --warmup 6 --opt wide=0 $opts >> $OUT/stream_on_image.jsonl 2>> $OUT/err.txt
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run36/stream_on_image.jsonl"):
    r = json.loads(ln); print(r["opts"], r["ms_mean"], r["ms_min"])
PY
