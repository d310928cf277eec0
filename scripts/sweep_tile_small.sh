for T in 0 1 2 3; do
python bench.py --no-cpu-baseline --no-companions --steps 300 --opt tile_small=$T 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('tile_small=$T headline', r['value'], r['roofline']['kernel_avg_ms'], r['roofline']['kernel_min_ms'])"
python scripts/run_query.py --config c2 --query closest --opt tile_small=$T 2>/dev/null | cut -c1-170
python scripts/run_query.py --config c4 --query closest --opt tile_small=$T 2>/dev/null | cut -c1-170
python scripts/run_query.py --config c5i --query any --opt tile_small=$T 2>/dev/null | cut -c1-170
python scripts/run_query.py --config c5i --res 2048 --query closest --steps 8 --opt tile=0 --opt tile_small=$T 2>/dev/null | cut -c1-190
done
