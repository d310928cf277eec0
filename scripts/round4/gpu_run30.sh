#!/bin/bash
# trips per wave (timeline build): VALU instructions per trip of the two headline kernels
OUT=gpurun_out/r04_run30
mkdir -p $OUT
export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/timeline/libtriro_hip.so
python scripts/exp_timeline.py --res 1024 --query closest --warmup 14 > $OUT/timeline_c5i.json 2> $OUT/err.txt
python scripts/exp_timeline.py --hash-rays 12500000 --query closest --warmup 6 > $OUT/timeline_c5s.json 2>> $OUT/err.txt
python scripts/exp_timeline.py --hash-rays 10000000 --mesh c2 --query any --warmup 6 > $OUT/timeline_c3.json 2>> $OUT/err.txt
python - <<'PY'
import json
for n in ("c5i", "c5s", "c3"):
    try:
        r = json.load(open(f"gpurun_out/r04_run30/timeline_{n}.json"))
        print(n, "event_ms", r["event_ms"], "waves", r["waves"], "wave_us", r["wave_us"], "trips", r.get("trips"), "us_per_trip", r.get("us_per_trip"), "occupied", r["occupied_frac_of_slots"])
        print("   top", r["top_waves"][:3])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $OUT/err.txt
