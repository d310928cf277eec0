#!/bin/bash
# artefact refresh at the last tree: per-config table, builder, c5ii line, emulated N-rank bounds (Python driver and native step), host time per step
OUT=gpurun_out/r06_final8
mkdir -p $OUT
PROFILE_ROUND=r06 timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; wc -l $OUT/configs.jsonl
timeout 300 python scripts/bench_build.py > $OUT/build.jsonl 2>> $OUT/configs.err
timeout 600 python bench.py --workload c5ii --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c5ii.json 2>> $OUT/configs.err; head -c 200 $OUT/bench_c5ii.json; echo
: > $OUT/emulate.jsonl
for extra in "--exchange native" "" "--arrival copy" "--workload c5ii --steps 10 --warmup 3" "--workload c5ii --steps 10 --warmup 3 --exchange native"; do
  arr="--arrival none"; case "$extra" in *arrival*) arr="";; esac
  timeout 600 python bench.py --emulate-world 8 $arr --records slot --steps 200 --warmup 50 $extra >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r06_final8/emulate.jsonl'):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], c.get("step_driver"), "share", c["dst_share"], "arr", c["arrival"], "| plain", e["plain_1gpu_ms_per_step"], "rank0", e["rank0_ms_per_step"],
          "peer", e["peer_trace_ms_per_step"], "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
timeout 300 python scripts/profile_pipeline_host.py 2>&1 | grep "host time" > $OUT/pipeline_host.txt; cat $OUT/pipeline_host.txt
