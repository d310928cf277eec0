#!/bin/bash
# round 5: randomised parity sweep at the last tree, five more seeds (1000 iterations)
OUT=gpurun_out/r05_33; mkdir -p $OUT
for S in 531 532 533 534 535; do
  timeout 1500 python scripts/fuzz_parity.py --iters 200 --seed $S > $OUT/fuzz_seed$S.txt 2>&1
  echo "seed $S rc=$?"; tail -1 $OUT/fuzz_seed$S.txt
done
