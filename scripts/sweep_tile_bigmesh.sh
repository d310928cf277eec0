#!/bin/bash
# 8x8 tiles vs rows at 4 M and 16.7 M rays as a function of triangles per ray (icosphere 7 / 8 / 9)
for S in 7 8 9; do for R in 2048 4096; do for T in 0 2; do
python scripts/run_query.py --config c5i --subdiv $S --res $R --query closest --steps 6 --opt tile=$T --opt tile_small=0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({'subdiv':$S,'tris':r['tris'],'rays':r['rays'],'tile':$T,'ms':r['ms_mean']}))"
done; done; done
