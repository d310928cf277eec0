#!/bin/bash
# streaming launch on coherent flat batches with small per-wave ranges, and on small incoherent batches
for R in 2048 4096; do
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 --flat --opt stream=0 2>/dev/null
for W in 64 128 256; do for M in 16 32 48; do
python scripts/run_query.py --config c5i --res $R --query closest --steps 8 --flat --opt stream=2 --opt stream_rays=$W --opt stream_refill=$M 2>/dev/null
done; done; done
for N in 262144 1048576 2097152; do for M in headline bunny; do
python scripts/run_hash.py --n $N --mesh $M --opt stream=0 2>/dev/null
for W in 64 128 256 512; do
python scripts/run_hash.py --n $N --mesh $M --opt stream=2 --opt stream_rays=$W 2>/dev/null
done; done; done
