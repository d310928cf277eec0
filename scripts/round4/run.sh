#!/bin/bash
# The GPU runs of round 4 as ONE parametrised runner (run on the box through gpurun, each step under its own timeout):
#   scripts/round4/run.sh suite                 full GPU suite, smoke, bench (default + the driver's arguments), rocprof summary r04m
#   scripts/round4/run.sh emulate [8]           emulated N-rank bounds: 4-byte / 12-byte records x default / even shards x arrival
#                                               none / copy, c5i weak + strong + c5ii (profiles/r04_emulate_records.jsonl)
#   scripts/round4/run.sh eight-ranks           bench.py --gpus 8 --backend gloo on one GPU, five modes (r04_eight_ranks_gloo.txt)
#   scripts/round4/run.sh stream-budget         streaming launch: refill threshold sweep, time + steps per wave (timeline build) +
#                                               SQ_INSTS_VALU (r04_stream_refill_valu.txt); builds the timeline variant first
#   scripts/round4/run.sh order-transfer        launches 1..6 on a fresh handle / after a change of resolution (r04_order_transfer_series.jsonl)
#   scripts/round4/run.sh stream-on-image       the streaming launch forced on the headline image, small ranges
#   scripts/round4/run.sh huge                  the opt-in 21 M / 84 M-triangle test
#   scripts/round4/run.sh fuzz [iters] [seed]   randomised parity sweep
set -u
WHAT=${1:-suite}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_$WHAT
mkdir -p $OUT
cd $REPO
show_emulation() {
python - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    r = json.loads(ln); e = r["emulation"]; c = r["config"]
    print(c["workload"][:11], r["emulated_world"], c["record_form"][:9], "share", c["dst_share"], "ch", c["chunks"], "arr", c["arrival"], "rays0", c["rays_rank0"],
          "| plain", e["plain_1gpu_ms_per_step"], "rank0", e["rank0_ms_per_step"], "own", e["rank0_own_trace_only_ms"], "peer", e["peer_trace_ms_per_step"],
          "expand", e["expansion_alone_ms"], e["expansion_GBps"], "| implied", e["implied_scaling_vs_1gpu"], "ok", r["verified"])
PY
}
case $WHAT in
suite)
  bash scripts/round4/gpu_final.sh ;;
emulate)
  N=${1:-8}; : > $OUT/emulate.jsonl
  for rec in slot packed; do
    for extra in "" "--dst-share 1" "--arrival copy" "--scaling strong" "--workload c5ii --steps 10 --warmup 3" "--workload c5ii --steps 10 --warmup 3 --arrival copy" "--workload c5ii --steps 10 --warmup 3 --chunks 0"; do
      arr="--arrival none"; case "$extra" in *arrival*) arr="";; esac
      timeout 600 python bench.py --emulate-world $N $arr --records $rec --steps 200 --warmup 50 $extra >> $OUT/emulate.jsonl 2>> $OUT/err.txt
    done
  done
  show_emulation $OUT/emulate.jsonl ;;
eight-ranks)
  : > $OUT/eight_ranks.txt
  for args in "" "--scaling strong" "--workload c5ii --total-rays 20000001" "--records packed" "--dst-share 1"; do
    timeout 600 python bench.py --gpus 8 --backend gloo --steps 3 --warmup 1 --min-warmup-ms 0 --no-cpu-baseline --no-companions $args > $OUT/line.json 2> $OUT/err.txt
    echo "rc=$? args=[$args]" >> $OUT/eight_ranks.txt
    python -c "
import json
r = json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); c = r['config']
print('  n_gpus', r['n_gpus'], r['scaling'], 'verified', r['verified'], 'rays_total', c['rays_total'], 'shard_rays', c['shard_rays'], 'dst_share', c['dst_share'])" >> $OUT/eight_ranks.txt 2>&1
  done
  cat $OUT/eight_ranks.txt ;;
stream-budget)
  make -C trimesh-ray-optix_amd/csrc OUTDIR=../lib_var/timeline EXTRA=-DTR_TIMELINE=131072 > /dev/null 2>&1
  : > $OUT/sweep.txt
  for rf in 8 16 24 32 48; do
    echo "== stream_refill=$rf" >> $OUT/sweep.txt
    timeout 300 python scripts/run_query.py --config c5s --query closest --steps 10 --warmup 3 --opt stream_refill=$rf --opt wide=0 >> $OUT/sweep.txt 2>> $OUT/err.txt
    TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/timeline/libtriro_hip.so timeout 300 python scripts/exp_timeline.py --hash-rays 12500000 --query closest --warmup 4 --opt stream_refill=$rf --opt wide=0 > $OUT/tl_$rf.json 2>> $OUT/err.txt
    python -c "
import json; r = json.load(open('$OUT/tl_$rf.json'))
print('timeline: loop iterations per wave', r['trips'], 'refills of the longest waves', [w['handovers'] for w in r['top_waves']][:6])" >> $OUT/sweep.txt
  done
  (cd /tmp && export TMPDIR=/tmp && for rf in 8 16 32 48; do
    D=$OUT/pmc_$rf; mkdir -p $D
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU --output-format csv -d $D -- python3 $REPO/scripts/run_query.py --config c5s --query closest --steps 4 --warmup 2 --opt stream_refill=$rf --opt wide=0 > $D/log.txt 2>&1
    python3 -c "
import glob, csv, collections
d = collections.defaultdict(list)
for f in glob.glob('$D/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_query_stream' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print('pmc stream_refill $rf', {k: round(sum(v) / len(v) / 1e6, 3) for k, v in sorted(d.items())})" >> $OUT/sweep.txt
  done)
  cat $OUT/sweep.txt ;;
order-transfer)
  timeout 1700 python scripts/round4/exp_order_transfer2.py "$@" > $OUT/order_transfer2.jsonl 2> $OUT/err.txt
  python -c "
import json
for ln in open('$OUT/order_transfer2.jsonl'):
    r = json.loads(ln); print(r['scene'], 'mode', r['order_transfer'], 'fresh', r['fresh_handle_small_launches_1_6_ms'], '| after change', r['big_after_14_small_launches_1_6_ms'], '| steady', r['steady_big_ms'])" ;;
stream-on-image)
  for opts in "" "--opt stream=2" "--opt stream=2 --opt stream_rays=128" "--opt stream=2 --opt stream_rays=64" "--opt stream=2 --opt stream_rays=64 --opt stream_refill=16"; do
    timeout 300 python scripts/run_query.py --config c5i --query closest --steps 20 --warmup 6 --opt wide=0 $opts 2>> $OUT/err.txt | tee -a $OUT/stream_on_image.jsonl
  done ;;
huge)
  TRIRO_TEST_HUGE=1 timeout 1500 python -m pytest tests/test_gpu_round2.py -q -p no:cacheprovider -k large_meshes 2>&1 | tee $OUT/pytest_huge.txt | tail -3 ;;
fuzz)
  timeout 1700 python scripts/fuzz_parity.py --iters ${1:-120} --seed ${2:-51} 2>&1 | tee $OUT/fuzz.txt | tail -2 ;;
*)
  echo "unknown run: $WHAT"; exit 2 ;;
esac
