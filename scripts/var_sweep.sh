#!/bin/bash
# usage: scripts/var_sweep.sh v1 v2 ... ; each v is a directory under trimesh-ray-optix_amd/lib_var
# (built with `make OUTDIR=../lib_var/<v> EXTRA=...`) or "default"
for v in "$@"; do
  if [ $v = default ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/$v/libtriro_hip.so; fi
  for a in "" "--res 2048" "--rays hash" "--res 512"; do
    python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v','$a',j['value'],j['roofline']['kernel_avg_ms'])"
  done
done
