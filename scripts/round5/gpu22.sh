#!/bin/bash
# round 5: randomised parity sweep at the final tree, three more seeds
OUT=gpurun_out/r05_22
mkdir -p $OUT
for S in 521 522 523; do
  timeout 1500 python scripts/fuzz_parity.py --iters 200 --seed $S > $OUT/fuzz_seed$S.txt 2>&1
  echo "seed $S rc=$?"; tail -1 $OUT/fuzz_seed$S.txt
done
