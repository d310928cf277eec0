#!/bin/bash
# where do the streaming launch's VALU instructions go?  refill threshold sweep: time, trips / refills per wave (timeline
# build), VALU instructions per launch (PMC pass)
OUT=$PWD/gpurun_out/r04_run31
mkdir -p $OUT
REPO=$PWD
for rf in 8 16 24 32 48; do
  echo "== stream_refill=$rf" >> $OUT/sweep.txt
  python scripts/run_query.py --config c5s --query closest --steps 10 --warmup 3 --opt stream_refill=$rf --opt wide=0 >> $OUT/sweep.txt 2>> $OUT/err.txt
  TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/timeline/libtriro_hip.so python scripts/exp_timeline.py --hash-rays 12500000 --query closest --warmup 4 --opt stream_refill=$rf --opt wide=0 > $OUT/tl_$rf.json 2>> $OUT/err.txt
  python - $OUT/tl_$rf.json >> $OUT/sweep.txt <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
hand = [w["handovers"] for w in r["top_waves"]]
print("timeline: event_ms", r["event_ms"], "trips/wave", r["trips"], "refills of the 12 longest waves", hand)
PY
done
cd /tmp && export TMPDIR=/tmp
for rf in 8 16 32 48; do
  D=$OUT/pmc_$rf; mkdir -p $D
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $D -- python3 $REPO/scripts/run_query.py --config c5s --query closest --steps 4 --warmup 2 --opt stream_refill=$rf --opt wide=0 > $D/log.txt 2>&1
  python3 - "$D" "$rf" >> $OUT/sweep.txt <<'PY'
import sys, glob, csv, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_query_stream' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print("pmc stream_refill", sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 3) for k, v in sorted(d.items())})
PY
done
cat $OUT/sweep.txt
