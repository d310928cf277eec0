#!/bin/bash
# One SQ counter pass per option set.  usage: scripts/pmc_quick.sh tag "args1" "args2" ...
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for A in "$@"; do
  i=$((i+1)); OUT=$REPO/gpurun_out/pmcq_${TAG}_$i; mkdir -p $OUT
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline $A > $OUT/log.txt 2>&1
  python3 - "$OUT" "$A" <<'PY'
import sys,glob,csv,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_query' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2],'=>',{k:round(sum(v)/len(v)/1e6,2) for k,v in sorted(d.items())})
PY
done
