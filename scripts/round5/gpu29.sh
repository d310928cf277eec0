#!/bin/bash
# round 5: the automatic launch policy against forced settings on the interior scene (VERDICT r04 "next" #6), the terrain and the headline mesh
OUT=gpurun_out/r05_29; mkdir -p $OUT
bash scripts/policy_matrix.sh room "320 640 1280" > $OUT/policy_room.txt 2>/dev/null
bash scripts/policy_matrix.sh terrain "640 1024" > $OUT/policy_terrain.txt 2>/dev/null
bash scripts/policy_matrix.sh c5i "512 1024" > $OUT/policy_c5i.txt 2>/dev/null
wc -l $OUT/policy_*.txt; cat $OUT/policy_room.txt
