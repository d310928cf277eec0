#!/bin/bash
# round 5: the root of the wide hierarchy visited at refill (VERDICT r04 "next" #5): parity, A/B against the tree without it, PMC passes
OUT=gpurun_out/r05_8
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_round5.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest.txt; tail -4 $OUT/pytest.txt
REPO=$(pwd)
for V in base noroot base noroot; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c5s --query closest --steps 12 --warmup 6" "--config c5s --query count --steps 12 --warmup 6" "--config c5s --query any --steps 12 --warmup 6" \
           "--config c5s --query closest --steps 12 --warmup 6 --subdiv 9" "--config c3 --query any --steps 12 --warmup 6 --opt wide=1" "--config c3 --query closest --steps 12 --warmup 6 --opt wide=1"; do
    python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['tris'], ' '.join(r['opts']), r['ms_mean'], r['ms_min'], r['mrays_per_s'])" >> $OUT/ab_root.txt
  done
done
cat $OUT/ab_root.txt
cd /tmp && export TMPDIR=/tmp
for V in base noroot; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    D=$REPO/$OUT/pmc_${V}_$(echo $P | cut -c1-7 | tr -d ' ')
    rocprofv3 --pmc $P --output-format csv -d $D -- python3 $REPO/scripts/run_query.py --config c5s --query closest --steps 4 --warmup 2 > $D.log 2>&1
  done
done
cd $REPO
python3 - <<'PY'
import glob,csv,collections,os
for V in ('base','noroot'):
    d=collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/r05_8/pmc_{V}_*/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_query_wide' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
    print(V, {k:round(sum(v)/len(v)/1e6,3) for k,v in sorted(d.items())}, 'L2 hit', round(sum(d['TCC_HIT_sum'])/max(1,sum(d['TCC_HIT_sum'])+sum(d['TCC_MISS_sum'])),3) if d.get('TCC_HIT_sum') else None)
PY
