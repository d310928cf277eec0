#!/bin/bash
OUT=gpurun_out/r04_run16
mkdir -p $OUT
timeout 600 python scripts/round4/exp_order_transfer.py > $OUT/order_transfer.json 2> $OUT/order_transfer.err; cat $OUT/order_transfer.json; tail -3 $OUT/order_transfer.err
