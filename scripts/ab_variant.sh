REPO=$(pwd)
for V in base steal7; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for k in 1 2; do
  python bench.py --no-cpu-baseline --no-companions --steps 300 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V headline', r['value'], r['roofline']['kernel_avg_ms'])"
  done
  python scripts/run_query.py --config c2 --query closest 2>/dev/null | cut -c1-160
  python scripts/run_query.py --config c4 --query closest 2>/dev/null | cut -c1-160
  python scripts/run_query.py --config c5i --query first 2>/dev/null | cut -c1-160
  python scripts/run_query.py --config c5i --res 2048 --query closest --flat --steps 8 2>/dev/null | cut -c1-160
  python scripts/run_hash.py --n 1048576 --mesh headline 2>/dev/null | cut -c1-160
done
