#!/bin/bash
OUT=gpurun_out/r05_25; mkdir -p $OUT
python scripts/round5/exp_moving_camera.py > $OUT/moving.json 2> $OUT/err.txt; cat $OUT/moving.json; tail -2 $OUT/err.txt
