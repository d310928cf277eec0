#!/bin/bash
# bound sharing between the lanes of a split ray: which donor threshold now?
for C in "--config c5i --query closest" "--config c4 --query closest" "--config c2 --query closest" "--config c5i --query any" "--config c5i --res 512 --query closest" "--config c5i --res 2048 --query closest" "--config c5i --query first"; do
  for A in "" "--opt steal=48" "--opt steal=32" "--opt steal=24" "--opt steal=16" "--opt steal=8"; do
    python scripts/run_query.py $C $A 2>&1 | tail -1
  done
done
