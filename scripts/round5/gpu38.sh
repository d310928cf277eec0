#!/bin/bash
# round 5: the emulation lines of gpu_final.sh once more (the c5ii rank-0 step read 7.5 ms in one run of the final script, 1.68 in the bisect)
OUT=gpurun_out/r05_49; mkdir -p $OUT; : > $OUT/emulate.jsonl
timeout 600 python bench.py --workload c5ii --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c5ii.json 2>> $OUT/err.txt
for extra in "" "--arrival copy" "--scaling strong" "--workload c5ii --steps 10 --warmup 3" "--workload c5ii --steps 10 --warmup 3 --arrival copy"; do
  arr="--arrival none"; case "$extra" in *arrival*) arr="";; esac
  timeout 600 python bench.py --emulate-world 8 $arr --records slot --steps 200 --warmup 50 $extra >> $OUT/emulate.jsonl 2>> $OUT/err.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05_49/emulate.jsonl'):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], r["emulated_world"], "share", c["dst_share"], "arr", c["arrival"], "| plain", e["plain_1gpu_ms_per_step"], "rank0", e["rank0_ms_per_step"],
          "peer", e["peer_trace_ms_per_step"], "expand", e["expansion_alone_ms"], "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
