#!/bin/bash
# a scene family the launch policy was NOT tuned on (open height field, grazing rays, sky misses): parity tests,
# then the automatic policy against forced alternatives
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_terrain.py -x -q 2>&1 | tail -3
for RES in 640 1024 1920; do
for O in "" "--opt split=0" "--opt tile=0 --opt tile_small=0" "--opt tile=2" "--opt grid_nodes=0" "--opt grid_nodes=2" "--opt steal=0" "--opt adaptive=0" "--opt split_outlier=0" "--opt occ8=2" ""; do
  timeout 120 python scripts/run_query.py --config terrain --query closest --res $RES --steps 100 --warmup 40 $O 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('terrain', r['rays'], '$O' or 'auto', r['ms_mean'], r['ms_min'])"
done
done
for Q in count any location; do
  timeout 120 python scripts/run_query.py --config terrain --query $Q --res 1024 --steps 30 --warmup 12 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print('terrain', r['rays'], '$Q auto', r['ms_mean'], r['ms_min'])"
done
