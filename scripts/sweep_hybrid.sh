#!/bin/bash
# hybrid footprint (rows for expensive regions, 8x8 tiles for cheap ones) on the 1 M-ray image configs
for H in 0 10 20 35 50 70 100; do
python bench.py --no-cpu-baseline --no-companions --steps 400 --opt hybrid=$H 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({'hybrid':$H,'config':'headline','value':r['value'],'kernel_avg_ms':r['roofline']['kernel_avg_ms'],'min':r['roofline']['kernel_min_ms']}))"
for A in "--config c2 --query closest" "--config c4 --query closest" "--config c5i --query any" "--config c5i --res 512 --query closest"; do
python scripts/run_query.py $A --steps 40 --warmup 12 --opt hybrid=$H 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({'hybrid':$H,'config':r['config'],'query':r['query'],'rays':r['rays'],'ms':r['ms_mean'],'min':r['ms_min']}))"
done; done
