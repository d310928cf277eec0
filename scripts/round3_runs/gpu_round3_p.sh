#!/bin/bash
Q="timeout 120 python scripts/run_query.py --steps 30 --warmup 8 --config c5s --query closest"
for O in "" "--presort 3" "--presort 3 --opt stream_dynamic=0" "--opt stream_dynamic=0" "--presort 4 --opt stream_dynamic=0" "--presort 4 --opt stream_dynamic=0 --opt stream_rays=512"; do
  $Q $O 2>&1 | grep -v amdgpu | cut -c1-220
done
