#!/bin/bash
# rocprofv3 summary of the headline run at the last tree (after the owner rule on exact ties): r06c
OUT=gpurun_out/r06_prof_c
mkdir -p $OUT
bash scripts/profile_bench.sh r06c --steps 1000 --warmup 50 --no-companions > $OUT/profile_bench.txt 2>&1
PROFILE_STEPS=1000 PROFILE_CMD="bench.py --steps 1000 --warmup 50 --no-companions --no-cpu-baseline (last tree of round 6)" python scripts/summarize_profile.py r06c > $OUT/r06c_summary.txt 2>&1
cp profiles/r06c_summary.* $OUT/ 2>/dev/null
sed -n '/Timed region/p;/## derived/,$p' $OUT/r06c_summary.md
