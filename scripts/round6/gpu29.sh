#!/bin/bash
mkdir -p gpurun_out; OUT=gpurun_out/r06_two_ranks29.txt; : > $OUT
for E in "" "RCCL_ENABLE_MULTIPLE_RANKS_PER_GPU=1" "NCCL_IGNORE_DUPLICATE_GPU=1"; do
  echo "== env: $E" >> $OUT
  env $E HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 scripts/round6/two_ranks_one_gpu.py 2>&1 | grep -v amdgpu.ids | grep -i "error\|duplicate\|ok\|invalid\|nccl" | head -12 | cut -c1-300 >> $OUT
done
cat $OUT
