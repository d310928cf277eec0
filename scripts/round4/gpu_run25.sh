#!/bin/bash
# ragged last tile row (direct launch + expansion), then the weighted weak-scaling emulation again
OUT=gpurun_out/r04_run25
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_round4.py -q -p no:cacheprovider -k "ragged_last_tile_row or slot_form or weak" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -15 $OUT/pytest.txt
: > $OUT/emulate.jsonl
for share in none 0.9 0.8 0.7 0.6 0.5; do
  extra=""; [ "$share" != "none" ] && extra="--dst-share $share"
  python bench.py --emulate-world 8 --arrival none $extra --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
done
for share in 0.7 0.6; do
  python bench.py --emulate-world 8 --arrival copy --dst-share $share --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  python bench.py --emulate-world 4 --arrival none --dst-share $share --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  python bench.py --emulate-world 2 --arrival none --dst-share $share --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run25/emulate.jsonl"):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(r["emulated_world"], "share", c["dst_share"], "arr", c["arrival"], "rays0", c["rays_rank0"], "peer", c["rays_peer"], "| plain", e["plain_1gpu_ms_per_step"],
          "rank0", e["rank0_ms_per_step"], "own", e["rank0_own_trace_only_ms"], "peer", e["peer_trace_ms_per_step"], "expand", e["expansion_alone_ms"],
          "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
