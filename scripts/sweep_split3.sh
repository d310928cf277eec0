#!/bin/bash
for rep in 1 2; do
for A in "--opt tile=2 --opt split=4" "--opt tile=2 --opt split=4 --opt split4=1" "--opt tile=2 --opt split=4 --opt steal=48" "--opt tile=2 --opt split=4 --opt steal=96" "--opt tile=2 --opt split=4 --opt split_steal=2" "--opt tile=2 --opt split=4 --opt xcd_chunk=2" "--opt tile=2 --opt split=4 --opt xcd_chunk=8"; do
  python scripts/run_query.py --config c5i --query closest $A 2>&1 | tail -1
done
done
for R in 512 720 1024 1448; do
  for A in "--flat" "--flat --opt split=4" "--flat --opt split=5" "--opt tile=2 --opt split=4" "--opt tile=2 --opt split=5"; do
    python scripts/run_query.py --config c5i --query closest --res $R $A 2>&1 | tail -1
  done
done
