#!/bin/bash
# round 5: three more boundaries of tr_policy re-checked at the final tree
OUT=gpurun_out/r05_44; mkdir -p $OUT; : > $OUT/ab.txt
# (1) TILE_MIN_RAYS = 4 M: 8x8 tiles on their own for large coherent batches
for RES in 2896 4096; do for O in "" "--opt tile=0 --opt tile_small=0" "--opt tile=2"; do
  python scripts/run_query.py --config c5i --query closest --res $RES --steps 20 --warmup 8 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('tiles c5i', r['rays'], 'closest', '$O' or 'auto', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
done; done
# (2) WIDE_LIST_MIN_TRIS = 500 k: the multi-hit list query on the 8-wide nodes (direct launch)
for C in room terrain c4 c2 soup; do for O in "--opt wide_direct=1" "--opt wide_direct=0" "--opt wide_direct=2"; do
  python scripts/run_query.py --config $C --query location --steps 40 --warmup 12 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('list $C', r['rays'], r['tris'], 'location', '$O', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
done; done
python scripts/run_query.py --config room --res 1280 --query location --steps 40 --warmup 12 --opt wide_direct=1 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('list room', r['rays'], r['tris'], 'location', 'wide_direct=1', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
python scripts/run_query.py --config room --res 1280 --query location --steps 40 --warmup 12 --opt wide_direct=0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('list room', r['rays'], r['tris'], 'location', 'wide_direct=0', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
# (3) WIDE_MIN_TRIS = 1 M: the streaming launch on the 8-wide nodes, on the meshes around the boundary (incoherent rays: the c5s-style hash rays are made for the headline mesh only -> terrain / room images are coherent; use c5s at subdivision 7 = 327 k triangles and 8 = 1.31 M)
for SD in 7 8; do for O in "--opt wide=0" "--opt wide=1"; do
  python scripts/run_query.py --config c5s --subdiv $SD --query closest --rays 6000000 --steps 16 --warmup 6 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('wide c5s', r['rays'], r['tris'], 'closest', '$O', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
  python scripts/run_query.py --config c5s --subdiv $SD --query any --rays 6000000 --steps 16 --warmup 6 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('wide c5s', r['rays'], r['tris'], 'any', '$O', r['ms_mean'], r['ms_min'])" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
