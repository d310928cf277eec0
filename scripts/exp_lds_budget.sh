#!/bin/bash
# What does a per-workgroup LDS table cost in occupancy alone?  The query kernels are built with
# TR_LDS_PAD = 4 / 8 / 16 KiB of extra (unused) LDS per 128-thread workgroup -- the footprint of the top
# 6 / 7 / 8 levels of the tree as 64-byte nodes -- and timed on the headline and on the large configs.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for P in 4096 8192 16384; do      # build the padded variants if they are not there
  [ -f $REPO/trimesh-ray-optix_amd/lib_var/ldspad$P/libtriro_hip.so ] || make -s -C $REPO/trimesh-ray-optix_amd/csrc -j4 OUTDIR=../lib_var/ldspad$P EXTRA=-DTR_LDS_PAD=$P > /dev/null
done
for P in 0 4096 8192 16384; do
  if [ $P = 0 ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/ldspad$P/libtriro_hip.so; fi
  for A in "--config c5i --query closest" "--config c5i --res 4096 --query closest --steps 8" "--config c4 --query count" "--config c2 --query closest"; do
    python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); r['lds_pad_bytes']=$P; print(json.dumps(r))"
  done
done
