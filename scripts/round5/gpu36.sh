#!/bin/bash
OUT=gpurun_out/r05_36; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -3 $OUT/pytest.txt
