#!/bin/bash
# round 5: rocprofv3 trace + PMC passes of the configs beside the headline at the final tree -> profiles/r05_{c3any,c3closest,c5s,c4count,c4loc}_summary.*
KERNEL_KEY=k_query_stream bash scripts/profile_query.sh r05_c3any --config c3 --query any
KERNEL_KEY=k_query_stream bash scripts/profile_query.sh r05_c3closest --config c3 --query closest
KERNEL_KEY=k_query_wide bash scripts/profile_query.sh r05_c5s --config c5s --query closest
KERNEL_KEY=k_query_direct bash scripts/profile_query.sh r05_c4count --config c4 --query count
KERNEL_KEY=k_query_direct bash scripts/profile_query.sh r05_c4loc --config c4 --query location
ls profiles | grep r05_c | head -20
mkdir -p gpurun_out/r05_15 && cp profiles/r05_c3any_summary.* profiles/r05_c3closest_summary.* profiles/r05_c5s_summary.* profiles/r05_c4count_summary.* profiles/r05_c4loc_summary.* gpurun_out/r05_15/
