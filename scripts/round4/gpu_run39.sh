#!/bin/bash
OUT=gpurun_out/r04_run39
mkdir -p $OUT
timeout 1500 python scripts/round4/ab_stream_pool.py "$@" > $OUT/ab_stream_pool.jsonl 2> $OUT/err.txt
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run39/ab_stream_pool.jsonl"):
    r = json.loads(ln)
    print(r["config"], "refill", r["stream_refill"], "min", r["pool_min"], "wait", r["pool_wait"], " ".join(f"{q}: {r[q]['ms']} ({r[q]['ratio']}) {'ok' if r[q]['same'] else 'DIFF'}" for q in ("closest", "first", "any", "count")))
PY
tail -3 $OUT/err.txt
