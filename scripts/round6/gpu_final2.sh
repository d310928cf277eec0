#!/bin/bash
# end-of-round validation: the whole GPU suite, smoke, the bench lines (default and the driver's arguments), the rocprof summary r06b,
# the other configs, the builder, the emulated N-rank bounds at the final tree
OUT=gpurun_out/r06_final2
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_full.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu_full.txt; tail -5 $OUT/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 330 $OUT/bench_default.json; echo
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench_default.err; head -c 330 $OUT/bench_driver_args.json; echo
bash scripts/profile_bench.sh r06b --steps 1000 --warmup 50 --no-companions > $OUT/profile_bench.txt 2>&1
PROFILE_STEPS=1000 PROFILE_CMD="bench.py --steps 1000 --warmup 50 --no-companions --no-cpu-baseline (tree of round 6)" python scripts/summarize_profile.py r06b > $OUT/r06b_summary.txt 2>&1
cp profiles/r06b_summary.* $OUT/ 2>/dev/null
PROFILE_ROUND=r06 timeout 900 python scripts/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; wc -l $OUT/configs.jsonl
timeout 300 python scripts/bench_build.py > $OUT/build.jsonl 2>> $OUT/configs.err
timeout 600 python bench.py --workload c5ii --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_c5ii.json 2>> $OUT/bench_default.err; head -c 200 $OUT/bench_c5ii.json; echo
: > $OUT/emulate.jsonl
for extra in "" "--arrival copy" "--scaling strong" "--workload c5ii --steps 10 --warmup 3" "--workload c5ii --steps 10 --warmup 3 --arrival copy"; do
  arr="--arrival none"; case "$extra" in *arrival*) arr="";; esac
  timeout 600 python bench.py --emulate-world 8 $arr --records slot --steps 200 --warmup 50 $extra >> $OUT/emulate.jsonl 2>> $OUT/emulate.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r06_final2/emulate.jsonl'):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], r["emulated_world"], "share", c["dst_share"], "arr", c["arrival"], "| plain", e["plain_1gpu_ms_per_step"], "rank0", e["rank0_ms_per_step"],
          "peer", e["peer_trace_ms_per_step"], "expand", e["expansion_alone_ms"], "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
# the host side of one pipelined step (python driver vs the native step) and the c5i weak run with the native step
timeout 300 python scripts/profile_pipeline_host.py > $OUT/pipeline_host.txt 2>&1; tail -2 $OUT/pipeline_host.txt
timeout 600 python bench.py --emulate-world 8 --arrival none --records slot --exchange native --steps 200 --warmup 50 > $OUT/emulate_native.json 2>> $OUT/emulate.err; head -c 300 $OUT/emulate_native.json; echo
