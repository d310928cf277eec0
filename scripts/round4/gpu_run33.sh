#!/bin/bash
OUT=gpurun_out/r04_run33
mkdir -p $OUT
timeout 1500 python scripts/round4/ab_stream_vote.py > $OUT/ab_stream_vote.jsonl 2> $OUT/err.txt
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run33/ab_stream_vote.jsonl"):
    r = json.loads(ln)
    print(r["config"], "refill", r["stream_refill"], "vote", r["leaf_vote"], " ".join(f"{q}: {r[q]['ms']} ({r[q]['ratio']}) {'ok' if r[q]['same'] else 'DIFF'}" for q in ("closest", "first", "any", "count")))
PY
tail -3 $OUT/err.txt
