#!/bin/bash
# round 4, GPU call 5: what bounds the expansion (calibration + PMC), watertight test, profiles of the wide launch
OUT=gpurun_out/r04_run5
mkdir -p $OUT
REPO=$PWD
timeout 600 python scripts/round4/expand_micro.py > $OUT/expand_micro.jsonl 2> $OUT/expand_micro.err
cat $OUT/expand_micro.jsonl
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    tag=$(echo $pmc | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $pmc --output-format csv -d $REPO/$OUT/pmc_m${m}_$tag -- python3 $REPO/scripts/round4/expand_micro.py --only $m --reps 5 > $REPO/$OUT/pmc_m${m}_$tag.log 2>&1
  done
done
cd $REPO
python3 - <<'PY'
import glob, csv, collections
for m in (0, 1):
    d = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r04_run5/pmc_m{m}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "expand" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("mode", m, {k: round(sum(v) / len(v), 1) for k, v in sorted(d.items())})
PY
timeout 900 python -m pytest tests/test_watertight.py -x -q -p no:cacheprovider > $OUT/pytest_watertight.txt 2>&1; tail -3 $OUT/pytest_watertight.txt
