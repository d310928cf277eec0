#!/bin/bash
# round 5: the GPU suite after the split of launch_query (launch_streaming)
OUT=gpurun_out/r05_35; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest.txt 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.txt
python scripts/run_query.py --config c3 --query any --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], r['ms_mean'], r['ms_min'])"
python scripts/run_query.py --config c5s --query closest --steps 12 --warmup 6 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], r['ms_mean'], r['ms_min'])"
python bench.py --steps 400 --warmup 50 --no-companions --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', r['value'], r['ms_per_step'], r['verified'])"
for A in "--config c4 --query count" "--config c4 --query location" "--config c2 --query closest" "--config terrain --query closest"; do python scripts/run_query.py $A --steps 60 --warmup 30 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config'], r['query'], r['ms_mean'], r['ms_min'])"; done
