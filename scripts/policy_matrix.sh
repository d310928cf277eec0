#!/bin/bash
# The automatic launch policy against forced alternatives on ONE scene family (a parametrised successor of
# scripts/round3_runs/gpu_round3_terrain.sh).  Closest hit at several image widths, then the other queries.
# usage (GPU box): scripts/policy_matrix.sh <config: terrain|soup|c2|c4|c5i|room> "<res list>" > profiles/<round>_policy_<config>.txt
CFG=${1:-terrain}
RESES=${2:-"640 1024 1920"}
for RES in $RESES; do
for O in "" "--opt split=0" "--opt tile=0 --opt tile_small=0" "--opt tile=2" "--opt grid_nodes=0" "--opt grid_nodes=2" "--opt steal=0" "--opt adaptive=0" "--opt split_outlier=0" "--opt sort_inline=0" "--opt wide_direct=3" ""; do
  timeout 180 python scripts/run_query.py --config $CFG --query closest --res $RES --steps 100 --warmup 40 $O 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$CFG', r['rays'], 'closest', '$O' or 'auto', r['ms_mean'], r['ms_min'])"
done
done
R1=$(echo $RESES | awk '{print $NF}'); [ "$CFG" = "terrain" ] && R1=1024
for Q in count any location; do
  for O in "" "--opt wide_direct=0" "--opt wide_direct=2" "--opt usteal=0"; do
    timeout 180 python scripts/run_query.py --config $CFG --query $Q --res $R1 --steps 30 --warmup 12 $O 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('$CFG', r['rays'], '$Q', '$O' or 'auto', r['ms_mean'], r['ms_min'])"
  done
done
