#!/bin/bash
# defaults after the 4-byte records: tests, expansion A/B, emulated bounds with the default share
OUT=gpurun_out/r04_run28
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -q -p no:cacheprovider -k "four_byte or two_ranks or emulated or ragged_last_tile_row or force_gather or bench_two_ranks or one_rank or slot_form" > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -6 $OUT/pytest.txt
TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/head/libtriro_hip.so python scripts/round4/ab_expand_libs.py --old-abi > $OUT/ab.txt 2>> $OUT/ab.err
python scripts/round4/ab_expand_libs.py >> $OUT/ab.txt 2>> $OUT/ab.err
cat $OUT/ab.txt
: > $OUT/emulate.jsonl
for rec in slot packed; do
  for N in 8 4 2; do
    python bench.py --emulate-world $N --arrival none --records $rec --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  done
  python bench.py --emulate-world 8 --arrival copy --records $rec --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  python bench.py --emulate-world 8 --arrival none --records $rec --dst-share 1 --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  python bench.py --emulate-world 8 --scaling strong --arrival none --records $rec --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  python bench.py --emulate-world 8 --workload c5ii --arrival none --records $rec --chunks 1 --steps 10 --warmup 3 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  python bench.py --emulate-world 8 --workload c5ii --arrival none --records $rec --steps 10 --warmup 3 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
  python bench.py --emulate-world 8 --workload c5ii --arrival copy --records $rec --chunks 1 --steps 10 --warmup 3 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run28/emulate.jsonl"):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], r["emulated_world"], c["record_form"][:9], "share", c["dst_share"], "ch", c["chunks"], "arr", c["arrival"], "rays0", c["rays_rank0"], "| plain", e["plain_1gpu_ms_per_step"],
          "rank0", e["rank0_ms_per_step"], "own", e["rank0_own_trace_only_ms"], "peer", e["peer_trace_ms_per_step"], "expand", e["expansion_alone_ms"], e["expansion_GBps"],
          "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
