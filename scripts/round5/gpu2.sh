#!/bin/bash
# round 5, second GPU run: the whole GPU suite on the fixed fused build, a short fuzz, occupancy variants
OUT=gpurun_out/r05_2
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
timeout 600 python scripts/fuzz_parity.py --iters 120 --seed 501 > $OUT/fuzz_501.txt 2>&1; tail -2 $OUT/fuzz_501.txt
rm -f $OUT/ab_*.txt
AB_SET=direct bash scripts/round5/ab.sh $OUT/ab_direct.txt base nofuse h7
AB_SET=stream bash scripts/round5/ab.sh $OUT/ab_stream.txt base nofuse sw6 sw7
cat $OUT/ab_direct.txt $OUT/ab_stream.txt
for k in 1 2; do python scripts/run_query.py --config c5i --query count --steps 40 --warmup 20; python scripts/run_query.py --config c4 --query count --steps 40 --warmup 20; done
TRIRO_HIP_LIBRARY=$PWD/trimesh-ray-optix_amd/lib_var/nofuse/libtriro_hip.so python scripts/run_query.py --config c5i --query count --steps 40 --warmup 20
