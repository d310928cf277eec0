#!/bin/bash
# A/B of the streaming launch (wave-level ray refill) on the incoherent configs.
for C in "c3 any" "c3 closest" "c3 count" "c5s closest" "c5s any"; do
  set -- $C
  python scripts/run_query.py --config $1 --query $2 --steps 8 --opt stream=0 2>/dev/null
  for R in 256 512 1024 2048; do for M in 8 16 32; do
    python scripts/run_query.py --config $1 --query $2 --steps 8 --opt stream=2 --opt stream_rays=$R --opt stream_refill=$M 2>/dev/null
  done; done
done
python scripts/run_query.py --config c5i --query closest --opt stream=0 2>/dev/null
python scripts/run_query.py --config c5i --query closest --opt stream=2 2>/dev/null
python scripts/run_query.py --config c5i --res 4096 --query closest --steps 8 --opt stream=0 2>/dev/null
python scripts/run_query.py --config c5i --res 4096 --query closest --steps 8 --opt stream=2 2>/dev/null
