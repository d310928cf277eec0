#!/bin/bash
# round 5: emulated N-rank bounds with the re-derived destination shares
export OUT=${OUT:-gpurun_out/r05_13}
mkdir -p $OUT
: > $OUT/emulate.jsonl
for extra in "" "--arrival copy" "--workload c5ii --steps 10 --warmup 3" "--workload c5ii --steps 10 --warmup 3 --arrival copy" "--records packed" "--workload c5ii --steps 10 --warmup 3 --records packed"; do
  arr="--arrival none"; case "$extra" in *arrival*) arr="";; esac
  rec="--records slot"; case "$extra" in *records*) rec="";; esac
  timeout 600 python bench.py --emulate-world 8 $arr $rec --steps 200 --warmup 50 $extra >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
done
for N in 2 4; do timeout 600 python bench.py --emulate-world $N --arrival none --records slot --steps 200 --warmup 50 >> $OUT/emulate.jsonl 2>> $OUT/emulate.err; done
python - <<'PY'
import json, os
for ln in open(os.environ.get('OUT', 'gpurun_out/r05_13') + '/emulate.jsonl'):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], r["emulated_world"], c["record_form"][:9], "share", c["dst_share"], "arr", c["arrival"], "| plain", e["plain_1gpu_ms_per_step"], "rank0", e["rank0_ms_per_step"],
          "peer", e["peer_trace_ms_per_step"], "expand", e["expansion_alone_ms"], "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
