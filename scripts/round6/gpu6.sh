#!/bin/bash
# round 6, GPU job 6: SQ counters of the headline launch, round 5's library against the current tree (same box); the emulation
# with the warm-up fixed, six processes
mkdir -p gpurun_out
export TRIRO_ABI_ANY=1
TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/r05/libtriro_hip.so bash scripts/pmc_quick.sh r06_r05 "--no-companions" > gpurun_out/r06_pmc_r05.txt 2>&1
unset TRIRO_HIP_LIBRARY
bash scripts/pmc_quick.sh r06_base "--no-companions" > gpurun_out/r06_pmc_base.txt 2>&1
unset TRIRO_ABI_ANY
cat gpurun_out/r06_pmc_r05.txt gpurun_out/r06_pmc_base.txt
OUT=gpurun_out/r06_emu2; mkdir -p $OUT; : > $OUT/log.txt
emu() { tag=$1; shift; timeout 600 python bench.py --emulate-world 8 --arrival none --records slot --workload c5ii --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=r['emulation']; print('$tag', 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], 'plain', e['plain_1gpu_ms_per_step'], 'mallocs', e['device_mallocs_in_timed_region'], 'implied', e['implied_scaling_vs_1gpu'], 'verified', r['verified'])" >> $OUT/log.txt; }
for i in 1 2 3 4 5 6; do emu run$i; done
timeout 600 python bench.py --emulate-world 8 --workload c5i --steps 40 --warmup 10 2>/dev/null | tail -1 > $OUT/c5i_weak.json
cat $OUT/log.txt; python -c "import json; r=json.load(open('$OUT/c5i_weak.json')); print('c5i weak', r['emulation'])"
