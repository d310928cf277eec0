#!/bin/bash
# option sweep on the headline config (1024^2 closest): are the defaults still the best?
for O in "" "--opt tile=2" "--opt steal=48" "--opt steal=96" "--opt steal=0" "--opt block_size=256" "--opt block_size=64" "--opt xcd_chunk=64" "--opt xcd_chunk=256" "--opt tile=2 --opt steal=32" "--opt compact=0"; do
python bench.py --no-cpu-baseline --no-companions --steps 300 $O 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps({'opts':'$O','value':r['value'],'kernel_avg_ms':r['roofline']['kernel_avg_ms'],'kernel_min_ms':r['roofline']['kernel_min_ms']}))"
done
