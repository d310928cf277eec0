#!/bin/bash
mkdir -p gpurun_out
Q="python scripts/run_query.py --steps 60 --warmup 24 --query closest"
(
for CFG in "room" "room --res 1280" "room --res 2560" "c5i --res 512" "c5i --res 768" "c5i --res 1024" "c5i --res 1448" "c4 --res 512" "c4" "c2 --res 512" "c2"; do
  $Q --config $CFG
  $Q --config $CFG --opt split_outlier=0
  $Q --config $CFG --opt split=3
done
) > gpurun_out/r3g_auto.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r3g_auto.jsonl | python3 -c "
import sys,json
for ln in sys.stdin:
    try: r=json.loads(ln)
    except Exception: print(ln[:200]); continue
    print(r['config'], r['rays'], ' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])
"
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r3g_bench.json 2> gpurun_out/r3g_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3g_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['kernel_avg_ms'], r['verified'], r['roofline']['cold_kernel_ms'], r['roofline']['moving_camera_kernel_ms'], json.dumps(r.get('ref_shape')['scenes']))
PY
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
