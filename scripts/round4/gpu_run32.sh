#!/bin/bash
# leaf block on every 2nd / 3rd / 4th trip of the streaming launch: time, steps per wave, VALU instructions
OUT=$PWD/gpurun_out/r04_run32
mkdir -p $OUT
REPO=$PWD
for v in base alt2 alt3; do
  LIB=$REPO/trimesh-ray-optix_amd/lib_var/$v/libtriro_hip.so; TL=$REPO/trimesh-ray-optix_amd/lib_var/${v}tl/libtriro_hip.so
  [ $v = base ] && LIB=$REPO/trimesh-ray-optix_amd/lib/libtriro_hip.so && TL=$REPO/trimesh-ray-optix_amd/lib_var/timeline/libtriro_hip.so
  echo "== $v" >> $OUT/sweep.txt
  for cfg in "c5s closest" "c3 any" "c3 closest"; do
    set -- $cfg
    TRIRO_HIP_LIBRARY=$LIB python scripts/run_query.py --config $1 --query $2 --steps 10 --warmup 3 --opt wide=0 >> $OUT/sweep.txt 2>> $OUT/err.txt
  done
  TRIRO_HIP_LIBRARY=$TL python scripts/exp_timeline.py --hash-rays 12500000 --query closest --warmup 4 --opt wide=0 > $OUT/tl_$v.json 2>> $OUT/err.txt
  python - $OUT/tl_$v.json >> $OUT/sweep.txt <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("timeline: event_ms", r["event_ms"], "loop iterations/wave", r["trips"], "refills of the longest waves", [w["handovers"] for w in r["top_waves"]][:6])
PY
done
cd /tmp && export TMPDIR=/tmp
for v in base alt2 alt3; do
  LIB=$REPO/trimesh-ray-optix_amd/lib_var/$v/libtriro_hip.so
  [ $v = base ] && LIB=$REPO/trimesh-ray-optix_amd/lib/libtriro_hip.so
  D=$OUT/pmc_$v; mkdir -p $D
  TRIRO_HIP_LIBRARY=$LIB rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU --output-format csv -d $D -- python3 $REPO/scripts/run_query.py --config c5s --query closest --steps 4 --warmup 2 --opt wide=0 > $D/log.txt 2>&1
  python3 - "$D" "$v" >> $OUT/sweep.txt <<'PY'
import sys, glob, csv, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_query_stream' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print("pmc", sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 3) for k, v in sorted(d.items())})
PY
done
cat $OUT/sweep.txt
