#!/bin/bash
# round 5: how the learned order converges over the first launches of a shape (fresh handles), sort_inline 1 / 0; the launch log of one handle
OUT=gpurun_out/r05_20
mkdir -p $OUT
SERIES_HANDLES=1 SERIES_N=24 TRIRO_DEBUG_LAUNCH=1 python scripts/round5/exp_launch_series.py "sort_inline=1" > $OUT/series_debug.jsonl 2> $OUT/debug.txt
grep -c "query 2" $OUT/debug.txt; grep "query 2" $OUT/debug.txt | tail -24 | cut -c1-200
python scripts/round5/exp_launch_series.py "sort_inline=1" "sort_inline=0" > $OUT/series.jsonl 2> $OUT/err.txt
python - <<'PY'
import json
for ln in open('gpurun_out/r05_20/series.jsonl'):
    r = json.loads(ln); print(r['opts'], 'first5', r['sum_first_5_ms'], 'mean 6-25', r['mean_6_25_ms'], 'last10', r['mean_last_10_ms']); print('  ', r['launch_ms_median_over_handles'])
PY
