#!/bin/bash
# direct launch: binary nodes (today's shapes) against the 8-wide nodes (wide_direct)
OUT=${1:-gpurun_out/r04_wide_direct}
mkdir -p $OUT
for cfg in "c4 count" "c4 location" "c4 closest" "c5i closest" "c5i count" "c5i first" "c2 closest" "c2 any" "terrain closest" "terrain count" "room closest" "room count"; do
  set -- $cfg
  for wd in 0 3; do
    echo "== $1 $2 wide_direct=$wd" >> $OUT/ab.txt
    timeout 300 python scripts/run_query.py --config $1 --query $2 --opt wide_direct=$wd --steps 40 --warmup 16 2>/dev/null | tail -1 >> $OUT/ab.txt
  done
done
python - $OUT/ab.txt <<'PY'
import sys, json
lines = open(sys.argv[1]).read().split("\n")
cur = None
for ln in lines:
    if ln.startswith("=="): cur = ln
    elif ln.startswith("{"):
        r = json.loads(ln); print(cur, r["ms_mean"], "ms (min", r["ms_min"], ")")
PY
