#!/bin/bash
# round 5: the whole GPU suite twice more on a fresh box (flakiness check), in the driver's form (-x -q)
OUT=gpurun_out/r05_30; mkdir -p $OUT
for k in 1 2; do
  timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest_$k.txt 2>&1; echo "run $k rc=$?"; tail -2 $OUT/pytest_$k.txt
done
