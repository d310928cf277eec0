#!/bin/bash
# round 6, GPU job 25: two more branch hints in the stealing kernels (the hand-over block unlikely, a backtrack that finds its
# far child in the LDS ring likely: the climb over parent links moves out of line) -- lib_var/hints against the shipped library
mkdir -p gpurun_out
AB_SET=direct timeout 1200 bash scripts/round5/ab.sh gpurun_out/r06_ab25.txt base hints base hints > gpurun_out/r06_ab25.log 2>&1
AB_SET=stream timeout 600 bash scripts/round5/ab.sh gpurun_out/r06_ab25s.txt base hints > gpurun_out/r06_ab25s.log 2>&1
cat gpurun_out/r06_ab25.txt gpurun_out/r06_ab25s.txt
