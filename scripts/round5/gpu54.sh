#!/bin/bash
# round 5: the c5ii emulation's rank-0 step reads 1.7-1.9 ms in short jobs and 7.5-8.8 ms at the end of long ones: first thing in a job, three times; then
# after a few minutes of other GPU work in the same job, three times
OUT=gpurun_out/r05_54; mkdir -p $OUT; : > $OUT/log.txt
emu() { timeout 600 python bench.py --emulate-world 8 --arrival none --records slot --workload c5ii --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=r['emulation']; print('$1', 'rank0', e['rank0_ms_per_step'], 'own', e['rank0_own_trace_only_ms'], 'peer', e['peer_trace_ms_per_step'], 'expand', e['expansion_alone_ms'], 'plain', e['plain_1gpu_ms_per_step'])" >> $OUT/log.txt; }
emu first; emu first; emu first
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" >> $OUT/log.txt
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_configs.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/log.txt
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" >> $OUT/log.txt
emu after; emu after; emu after
for i in 1 2 3 4 5 6; do python scripts/run_query.py --config c5s --query closest --steps 30 --warmup 5 > /dev/null 2>&1; done
emu after_more
cat $OUT/log.txt
