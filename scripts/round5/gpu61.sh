#!/bin/bash
# round 5: the very last checks: the emulation lines first (fresh job), then 200 fuzz iterations
OUT=gpurun_out/r05_61; mkdir -p $OUT; : > $OUT/emulate.jsonl
for extra in "" "--arrival copy" "--workload c5ii --steps 10 --warmup 3" "--workload c5ii --steps 10 --warmup 3 --arrival copy"; do
  arr="--arrival none"; case "$extra" in *arrival*) arr="";; esac
  timeout 600 python bench.py --emulate-world 8 $arr --records slot --steps 200 --warmup 50 $extra >> $OUT/emulate.jsonl 2>> $OUT/err.txt
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05_61/emulate.jsonl'):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], "share", c["dst_share"], "arr", c["arrival"], "| plain", e["plain_1gpu_ms_per_step"], "rank0", e["rank0_ms_per_step"], "own", e["rank0_own_trace_only_ms"],
          "peer", e["peer_trace_ms_per_step"], "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
timeout 1500 python scripts/fuzz_parity.py --iters 200 --seed 561 > $OUT/fuzz_seed561.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_seed561.txt
