#!/bin/bash
# round 5: where does the UNLEARNED launch lose?  (VERDICT r04 "next" #4)  cold launches (adaptive = 0) across the options that
# shape them, and the per-wave timeline of the cold and the steady launch
OUT=gpurun_out/r05_5
mkdir -p $OUT
Q="python scripts/run_query.py --config c5i --query closest --steps 40 --warmup 10"
for O in "" "--opt adaptive=0" "--opt adaptive=0 --opt tile=2" "--opt adaptive=0 --opt steal=48" "--opt adaptive=0 --opt steal=32" "--opt adaptive=0 --opt steal=16" \
         "--opt adaptive=0 --opt steal=32 --opt tile=2" "--opt adaptive=0 --opt steal=16 --opt tile=2" "--opt adaptive=0 --opt scramble=0" \
         "--opt adaptive=0 --opt tile_small=1" "--opt adaptive=0 --opt tile_small=2" "--opt adaptive=0 --opt grid_nodes=2" "--opt adaptive=0 --opt grid_nodes=2 --opt steal=32" \
         "--opt adaptive=0 --opt grid_nodes=2 --opt tile=2 --opt steal=16" "--opt adaptive=0 --opt xcd_chunk=0" "--opt adaptive=0 --opt xcd_chunk=16"; do
  $Q $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(' '.join(r['opts']) or 'default', r['ms_mean'], r['ms_min'])" >> $OUT/cold_options.txt
done
cat $OUT/cold_options.txt
export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/timeline/libtriro_hip.so
for O in "" "--opt adaptive=0 --opt grid_nodes=2" "--opt adaptive=0 --opt grid_nodes=2 --opt tile=2" "--opt adaptive=0 --opt grid_nodes=2 --opt steal=16"; do
  python scripts/exp_timeline.py --waves-per-simd 6 --warmup 20 $O >> $OUT/timeline.jsonl 2>> $OUT/timeline.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05_5/timeline.jsonl'):
    j=json.loads(ln)
    print(j['opts'], j['event_ms'], 'waves', j['waves'], 'span', j['span_us'], 'wave_us', j['wave_us'], 'last_start', j['last_start_us'], 'occupied', j['occupied_frac_of_slots'], 'tail', j['tail_waves'])
    print('   resident/10us', j['resident_waves_per_10us'])
    print('   top', [(w['us'], w['start'], w['trips']) for w in j['top_waves'][:8]])
PY
