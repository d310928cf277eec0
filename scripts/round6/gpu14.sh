#!/bin/bash
# round 6, GPU job 14: fewer levels of divergent control flow in the leaf block -- flat2 = the float32 inside test with two
# nested exits instead of four (-DTR_TRI_FLAT=2), foldb = the fold of a hit into the closest result as one mask and three
# selects (-DTR_FOLD_FLAT=1), both; against the shipped library (base), same box
mkdir -p gpurun_out
AB_SET=direct timeout 1500 bash scripts/round5/ab.sh gpurun_out/r06_ab14.txt base flat2 foldb flat2foldb base flat2foldb > gpurun_out/r06_ab14.log 2>&1
AB_SET=stream timeout 900 bash scripts/round5/ab.sh gpurun_out/r06_ab14s.txt base flat2foldb > gpurun_out/r06_ab14s.log 2>&1
cat gpurun_out/r06_ab14.txt gpurun_out/r06_ab14s.txt
