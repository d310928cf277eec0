#!/bin/bash
mkdir -p gpurun_out
Q="python scripts/run_query.py --config c5i --query closest --steps 30 --warmup 12"
(
for RES in 1024 2048 4096; do
  $Q --res $RES
  for SR in 256 512 1024; do for RF in 16 32; do
    $Q --res $RES --opt stream=2 --opt stream_tile=1 --opt stream_rays=$SR --opt stream_refill=$RF
  done; done
  $Q --res $RES --opt stream=2 --opt stream_tile=1 --opt stream_dynamic=0
  $Q --res $RES --opt stream=2
done
) > gpurun_out/r3c_stream_tile.jsonl 2>&1
cat gpurun_out/r3c_stream_tile.jsonl | grep -v amdgpu.ids | cut -c1-200
(
for S in 1 2 3; do for SS in 8 2; do
  $Q --res 512 --opt split=$S --opt split_steal=$SS
  python scripts/run_query.py --config c5i --query closest --steps 30 --warmup 12 --res 640 --opt split=$S --opt split_steal=$SS
done; done
) > gpurun_out/r3c_small_split.jsonl 2>&1
cat gpurun_out/r3c_small_split.jsonl | grep -v amdgpu.ids | cut -c1-200
python scripts/bench_configs.py > gpurun_out/r3c_configs.jsonl 2>/dev/null; cat gpurun_out/r3c_configs.jsonl | cut -c1-160
timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "rccl or packed" 2>&1 | tail -3
