#!/bin/bash
# round 6, GPU job 20: the unlearned launch (adaptive = 0) with waves DROPPING what they still owe G = 8 / 24 trips after
# the launch's last block has started (-DTR_SUSPEND_EXP=G: timing only, wrong results) -- how short does the first launch
# of a suspend / resume scheme get, and how many subtree entries would it have to write?  Beside it the same timeline
# build without the experiment.
mkdir -p gpurun_out; OUT=gpurun_out/r06_susp20.jsonl; : > $OUT
for V in timeline susp8 susp24; do
  export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so
  for M in c5i c4; do
    echo "# $V $M adaptive=0" >> $OUT
    python scripts/exp_timeline.py --mesh $M --opt adaptive=0 --waves-per-simd 6 >> $OUT 2>/dev/null
  done
done
python - <<'PY'
import json
for ln in open('gpurun_out/r06_susp20.jsonl'):
    if ln.startswith('#'): print(ln.strip()); continue
    r=json.loads(ln)
    print(' event_ms',r['event_ms'],'span',r['span_us'],'last_start',r['last_start_us'],'wave_us',r['wave_us'],'susp',r.get('suspended'))
    print(' resident',r['resident_waves_per_10us'])
PY
