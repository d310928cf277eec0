#!/bin/bash
# where does a flat tile (2x32 / 4x16) beat rows for small image-shaped closest-hit batches?  triangle density sweep
for S in 5 6 7 8; do for R in 512 1024; do for T in 0 1 2; do
python scripts/run_query.py --config c5i --subdiv $S --res $R --query closest --opt tile_small=$T 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({'subdiv':$S,'tris':r['tris'],'rays':r['rays'],'tile_small':$T,'ms':r['ms_mean']}))"
done; done; done
