#!/bin/bash
OUT=gpurun_out/r04_run27
mkdir -p $OUT
for k in 1 2; do
  TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/head/libtriro_hip.so python scripts/round4/ab_expand_libs.py --old-abi >> $OUT/ab.txt 2>> $OUT/ab.err
  python scripts/round4/ab_expand_libs.py >> $OUT/ab.txt 2>> $OUT/ab.err
done
cat $OUT/ab.txt
timeout 600 python -m pytest tests/test_gpu_round4.py -q -p no:cacheprovider -k "ragged_last_tile_row or four_byte" 2>&1 | tail -3
