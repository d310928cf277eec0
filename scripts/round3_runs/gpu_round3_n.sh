#!/bin/bash
export KERNEL_KEY=k_query_stream
timeout 500 bash scripts/profile_query.sh r03_c5s --config c5s --query closest > /dev/null 2>&1
timeout 500 bash scripts/profile_query.sh r03_c5s_presorted --config c5s --query closest --presort 3 > /dev/null 2>&1
for t in r03_c5s r03_c5s_presorted; do tail -22 gpurun_out/prof_$t/summary.txt; cp profiles/${t}_summary.* gpurun_out/ 2>/dev/null; done
ls profiles | grep r03_c5s
