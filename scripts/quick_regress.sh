python bench.py --no-cpu-baseline --no-companions 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('headline', r['value'], r['roofline']['kernel_avg_ms'])"
python scripts/run_query.py --config c2 --query closest 2>/dev/null
python scripts/run_query.py --config c4 --query closest 2>/dev/null
python scripts/run_query.py --config c4 --query count 2>/dev/null
python scripts/run_query.py --config c4 --query location 2>/dev/null
python scripts/run_query.py --config c5i --query any 2>/dev/null
python scripts/run_query.py --config c5i --query first 2>/dev/null
python scripts/run_query.py --config c3 --query any --steps 8 2>/dev/null
python scripts/run_query.py --config c3 --query closest --steps 8 2>/dev/null
python scripts/run_query.py --config c5s --query closest --steps 8 2>/dev/null
python scripts/run_query.py --config c5i --res 2048 --query closest --steps 8 2>/dev/null
python scripts/run_query.py --config c5i --res 4096 --query closest --steps 8 2>/dev/null
