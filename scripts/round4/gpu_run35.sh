#!/bin/bash
OUT=gpurun_out/r04_run35
mkdir -p $OUT
timeout 1700 python scripts/round4/exp_order_transfer2.py > $OUT/order_transfer2.jsonl 2> $OUT/err.txt
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run35/order_transfer2.jsonl"):
    r = json.loads(ln)
    print(r["scene"], "mode", r["order_transfer"], "fresh", r["fresh_handle_small_launches_1_6_ms"], "| after change", r["big_after_14_small_launches_1_6_ms"], "| steady", r["steady_big_ms"], r["steady_big_all"], "split", r["split_blocks"])
PY
tail -3 $OUT/err.txt
timeout 600 python -m pytest tests/test_gpu_round4.py -q -p no:cacheprovider -k "resolution" 2>&1 | tail -2
