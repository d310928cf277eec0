#!/bin/bash
OUT=$PWD/gpurun_out/r04_run34
mkdir -p $OUT
REPO=$PWD
timeout 1500 python scripts/round4/ab_stream_vote.py --votes 4,8,16 --configs c5s,c3 > $OUT/ab_stream_vote.jsonl 2> $OUT/err.txt
python - <<'PY'
import json
for ln in open("gpurun_out/r04_run34/ab_stream_vote.jsonl"):
    r = json.loads(ln)
    print(r["config"], "refill", r["stream_refill"], "vote", r["leaf_vote"], " ".join(f"{q}: {r[q]['ms']} ({r[q]['ratio']}) {'ok' if r[q]['same'] else 'DIFF'}" for q in ("closest", "first", "any", "count")))
PY
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  D=$OUT/pmc_$v; mkdir -p $D
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS --output-format csv -d $D -- python3 $REPO/scripts/run_query.py --config c5s --query closest --steps 4 --warmup 2 --opt wide=0 --opt stream_vote=$v --opt stream_leaf_vote=8 > $D/log.txt 2>&1
  python3 - "$D" "$v" <<'PY'
import sys, glob, csv, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_query_stream' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print("pmc stream_vote", sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 3) for k, v in sorted(d.items())})
PY
done
