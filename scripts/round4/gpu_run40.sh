#!/bin/bash
OUT=$PWD/gpurun_out/r04_run40
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for q in closest count; do
for v in 0 1; do
  D=$OUT/pmc_${q}_$v; mkdir -p $D
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $D -- python3 $REPO/scripts/run_query.py --config c5s --query $q --steps 4 --warmup 2 --opt wide=0 --opt stream_pool=$v --opt stream_pool_min=32 > $D/log.txt 2>&1
  python3 - "$D" "$q pool=$v" <<'PY'
import sys, glob, csv, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_query_stream' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print("pmc", sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 3) for k, v in sorted(d.items())})
PY
done
done
