#!/bin/bash
mkdir -p gpurun_out
Q="timeout 120 python scripts/run_query.py --steps 30 --warmup 8"
run() {
for C in "c5s closest" "c3 closest" "c3 any" "c3 count" "c3 first" "c5s any"; do set -- $C; $Q --config $1 --query $2; done
}
(echo '"A: 8 waves/SIMD requested"'; run; export TRIRO_HIP_LIBRARY=$GRAFT_REPO_ROOT/trimesh-ray-optix_amd/lib_var/libtriro_hip.so; echo '"B: baseline (7)"'; run; unset TRIRO_HIP_LIBRARY; echo '"A again"'; run) 2>&1 | grep -v amdgpu > gpurun_out/r3o_stream_occ.jsonl
python3 -c "
import json
for ln in open('gpurun_out/r3o_stream_occ.jsonl'):
    try: r=json.loads(ln)
    except Exception: print(ln[:100]); continue
    if isinstance(r,str): print(r); continue
    print(r['config'], r['query'], r['ms_mean'], r['ms_min'])
"
timeout 120 python scripts/run_query.py --steps 30 --warmup 12 --config c4 --query count --opt block_size=64 | grep -v amdgpu | cut -c1-200
timeout 120 python scripts/run_query.py --steps 30 --warmup 12 --config c4 --query count | grep -v amdgpu | cut -c1-200
