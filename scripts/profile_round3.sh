#!/bin/bash
# Round-3 evidence run (GPU box): per-config kernel trace + PMC passes, the literal default bench command
# under the kernel tracer, its PMC passes, the bench lines, the other configs.  Everything under its own
# timeout; summaries are copied to gpurun_out/ (merged back by gpurun), from where they go to profiles/.
mkdir -p gpurun_out/r03
prof() {  # tag kernel_key run_query args...
  local tag=$1 key=$2; shift 2
  KERNEL_KEY=$key timeout 400 bash scripts/profile_query.sh $tag "$@" > /dev/null 2>&1
  cp profiles/${tag}_summary.* gpurun_out/r03/ 2>/dev/null
  echo "== $tag"; tail -12 gpurun_out/prof_$tag/summary.txt | head -11
}
prof r03_c2closest k_query_direct --config c2 --query closest
prof r03_c3any k_query_stream --config c3 --query any
prof r03_c3closest k_query_stream --config c3 --query closest
prof r03_c4closest k_query_direct --config c4 --query closest
prof r03_c4count k_query_direct --config c4 --query count
prof r03_c4location k_query_direct --config c4 --query location
prof r03_roomclosest k_query_direct --config room --query closest
prof r03_roomcount k_query_direct --config room --query count
prof r03_c5s k_query_stream --config c5s --query closest
prof r03_terrainclosest k_query_direct --config terrain --query closest --res 1024
prof r03_terraincount k_query_direct --config terrain --query count --res 1024
# the headline: kernel trace of the literal default command, then the PMC passes of the same kernel
timeout 600 bash scripts/profile_default.sh r03m > /dev/null 2>&1
PROFILE_CMD="bench.py (default flags: 1000 timed steps)" PROFILE_STEPS=1000 python3 scripts/summarize_profile.py r03m k_query_direct > /dev/null 2>&1; cp profiles/r03m_summary.* gpurun_out/r03/
timeout 900 bash scripts/profile_bench.sh r03n --no-companions > /dev/null 2>&1
python3 scripts/summarize_profile.py r03n k_query_direct > /dev/null 2>&1; cp profiles/r03n_summary.* gpurun_out/r03/
cp profiles/r03n_summary.json profiles/r03_c5i_summary.json 2>/dev/null
# bench lines and the other configs (bench_configs.py reads the traffic figures from the summaries above)
timeout 600 python bench.py > gpurun_out/r03/r03_bench.json 2> gpurun_out/r03/r03_bench.err; echo "bench rc=$?"
timeout 600 python bench.py --workload c5ii --steps 20 --warmup 3 > gpurun_out/r03/r03_bench_c5ii.json 2> gpurun_out/r03/r03_bench_c5ii.err; echo "bench c5ii rc=$?"
timeout 900 python scripts/bench_configs.py > gpurun_out/r03/r03_configs.jsonl 2> /dev/null; echo "configs rc=$?"
timeout 300 python scripts/bench_build.py > gpurun_out/r03/r03_build.jsonl 2>/dev/null
cut -c1-330 gpurun_out/r03/r03_configs.jsonl
python3 - <<'PY'
import json
for f in ('gpurun_out/r03/r03_bench.json','gpurun_out/r03/r03_bench_c5ii.json'):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); rl=r['roofline']
        print(f, r['value'], r['ms_per_step'], rl['kernel_avg_ms'], rl['frac'], rl.get('traffic'), r.get('verified'), rl.get('cold_kernel_ms'), rl.get('first_launch_kernel_ms'), rl.get('moving_camera_kernel_ms'))
    except Exception as e: print(f, 'parse fail', e)
PY
