#!/bin/bash
# round 6, GPU job 5: (1) the bistable pipelined step of the c5ii emulation (VERDICT r05 "next" #2): first thing in the job and
# after five minutes of other processes, each once under rocprofv3 --kernel-trace, then variants in the "after" state;
# (2) in between: A/B of the current tree against round 5's library
OUT=gpurun_out/r06_emu; mkdir -p $OUT; : > $OUT/log.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
emu() { tag=$1; shift; timeout 600 python bench.py --emulate-world 8 --arrival none --records slot --workload c5ii --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=r['emulation']; print('$tag', 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], 'plain', e['plain_1gpu_ms_per_step'], 'verified', r['verified'])" >> $OUT/log.txt; }
emu first
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_first -- python3 bench.py --emulate-world 8 --arrival none --records slot --workload c5ii --steps 10 --warmup 3 > $OUT/prof_first.json 2> $OUT/prof_first.err
emu first2
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" >> $OUT/log.txt
export TRIRO_ABI_ANY=1
timeout 1200 bash scripts/round5/ab.sh gpurun_out/r06_ab5.txt r05 base > gpurun_out/r06_ab5.log 2>&1
unset TRIRO_ABI_ANY; unset TRIRO_HIP_LIBRARY
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" >> $OUT/log.txt
emu after
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_after -- python3 bench.py --emulate-world 8 --arrival none --records slot --workload c5ii --steps 10 --warmup 3 > $OUT/prof_after.json 2> $OUT/prof_after.err
emu after2
emu after_tracepri --trace-priority
emu after_arrpri --arrival-priority
GPU_MAX_HW_QUEUES=8 emu after_q8
GPU_MAX_HW_QUEUES=2 emu after_q2
emu after_chunks4 --chunks 4
emu after3
# keep the traces small: kernel name, queue, start, end
for d in prof_first prof_after; do f=$(find $OUT/$d -name "*kernel_trace.csv" | head -1); if [ -n "$f" ]; then python3 - "$f" > $OUT/$d.trace.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keys=rows[0].keys() if rows else []
print('#', list(keys))
t0=min(int(r['Start_Timestamp']) for r in rows)
for r in rows[-400:]:
    print(r.get('Queue_Id'), r.get('Stream_Id',''), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'][:70], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
PY
rm -rf $OUT/$d; fi; done
cat $OUT/log.txt; cat gpurun_out/r06_ab5.txt
