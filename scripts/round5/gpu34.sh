#!/bin/bash
# round 5: does a more generous split set help the moving camera?  (split_outlier = 0: every block the grid has room for; split = 3 / 2: room for 2 x / 4 x as many)
OUT=gpurun_out/r05_34; mkdir -p $OUT; : > $OUT/moving.jsonl
for O in "" "split_outlier=0" "split=3" "split=3 split_outlier=0" "split=2" "split_outlier=4" "split_steal=4" "steal=48"; do
  timeout 300 python scripts/round5/exp_moving_camera.py $O >> $OUT/moving.jsonl 2>> $OUT/err.txt
done
python - <<'PY'
import json
for l in open('gpurun_out/r05_34/moving.jsonl'):
    r=json.loads(l); print(r['opts'] or 'auto', 'same', r['same_tensors_ms'], 'moving', r['moving_camera_ping_pong_ms'], 'still at frame 3', r['moving_camera_frame3_only_ms'])
PY
