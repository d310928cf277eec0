#!/bin/bash
# streaming launch: ranges from a work counter (stream_dynamic=1) vs one static range per wave
for C in "--config c3 --query any" "--config c3 --query closest" "--config c3 --query count" "--config c5s --query closest"; do
  for A in "--opt stream_dynamic=0" "" "--opt stream_rays=256" "--opt stream_rays=128" "--opt stream_rays=1024" "--opt stream_rays=256 --opt stream_refill=24" "--opt stream_rays=256 --opt stream_refill=40"; do
    python scripts/run_query.py $C --steps 10 $A 2>&1 | tail -1
  done
done
