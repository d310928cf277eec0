#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_interior.py -x -q -m gpu 2>&1 | tail -15
Q="python scripts/run_query.py --steps 40 --warmup 16 --query closest"
(
$Q --config room
$Q --config room --opt split=2
$Q --config room --opt split=3
$Q --config room --opt split=4
$Q --config room --opt split=2 --opt split_steal=2
$Q --config room --opt grid_nodes=0
$Q --config room --opt grid_nodes=2
$Q --config room --opt tile_small=0 --opt split=0
$Q --config room --opt tile=2
$Q --config room --opt block_size=64
$Q --config room --flat
$Q --config room --flat --opt split=2
$Q --config room --res 1280
$Q --config room --res 2560
$Q --config c5i --res 512
$Q --config c5i --res 640
$Q --config c5i --res 768
$Q --config c5i --res 768 --opt split=3
$Q --config c5i --res 1024 --opt split=3
$Q --config c5i --res 1024
$Q --config c5i --res 512 --flat
$Q --config c5i --res 512 --flat --opt split=2
$Q --config c5i --res 512 --flat --opt split=3
$Q --config c2 --res 512
$Q --config c2 --res 512 --opt split=2
$Q --config c4 --res 512
$Q --config c4 --res 512 --opt split=4
python scripts/run_query.py --steps 40 --warmup 16 --config room --query count
python scripts/run_query.py --steps 40 --warmup 16 --config room --query location
python scripts/run_query.py --steps 40 --warmup 16 --config room --query any
) > gpurun_out/r3d_room.jsonl 2>&1
grep -v amdgpu.ids gpurun_out/r3d_room.jsonl | cut -c1-230
timeout 900 python bench.py > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r3d_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['kernel_avg_ms'], r['verified'], json.dumps(r.get('ref_shape'))[:900])
PY
