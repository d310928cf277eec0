#!/bin/bash
# round 5 EXPERIMENT: refit in one launch (agent-scope atomics across the XCDs) against the level-by-level pass: same bits? how fast?
OUT=gpurun_out/r05_31; mkdir -p $OUT; : > $OUT/refit.jsonl
for M in c5i c4 c2 deep; do
  timeout 300 python scripts/round5/exp_refit_single.py $M 12 >> $OUT/refit.jsonl 2>> $OUT/err.txt
  TRIRO_REFIT_SINGLE=1 timeout 300 python scripts/round5/exp_refit_single.py $M 12 >> $OUT/refit.jsonl 2>> $OUT/err.txt
done
for k in 1 2 3 4 5; do TRIRO_REFIT_SINGLE=1 timeout 300 python scripts/round5/exp_refit_single.py c5i 30 >> $OUT/refit.jsonl 2>> $OUT/err.txt; done
python - <<'PY'
import json, collections
rows=[json.loads(l) for l in open('gpurun_out/r05_31/refit.jsonl')]
ref={}
for r in rows:
    if not r['single']: ref[r['mesh']]=r['digests']
for r in rows:
    n=min(len(r['digests']),len(ref[r['mesh']]))
    print(r['mesh'], 'single' if r['single'] else 'levels', 'tris', r['tris'], 'depth', r['depth'], 'refit ms', r['refit_ms_median'], r['refit_ms_min'], 'same bits as the level pass:', r['digests'][:n]==ref[r['mesh']][:n])
PY
tail -3 $OUT/err.txt
