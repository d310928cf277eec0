#!/bin/bash
# 4-byte slot records: unit + two-rank + emulation tests, then the emulated bounds of both record forms
OUT=gpurun_out/r04_run26
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -q -p no:cacheprovider -k "four_byte or two_ranks or emulated or ragged_last_tile_row or force_gather or bench_two_ranks or one_rank" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -15 $OUT/pytest.txt
: > $OUT/emulate.jsonl
for rec in slot packed; do
  for share in none 0.7 0.6 0.5 0.4; do
    extra=""; [ "$share" != "none" ] && extra="--dst-share $share"
    python bench.py --emulate-world 8 --arrival none --records $rec $extra --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  done
  python bench.py --emulate-world 8 --arrival copy --records $rec --dst-share 0.5 --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  for share in 0.33 0.25; do
    python bench.py --emulate-world 8 --workload c5ii --arrival none --records $rec --dst-share $share --chunks 1 --steps 10 --warmup 3 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  done
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run26/emulate.jsonl"):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], r["emulated_world"], c["record_form"][:9], "share", c["dst_share"], "arr", c["arrival"], "rays0", c["rays_rank0"], "| plain", e["plain_1gpu_ms_per_step"],
          "rank0", e["rank0_ms_per_step"], "own", e["rank0_own_trace_only_ms"], "peer", e["peer_trace_ms_per_step"], "expand", e["expansion_alone_ms"], e["expansion_GBps"],
          "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
