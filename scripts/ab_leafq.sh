REPO=$(pwd)
for V in base leafq4 leafq8; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c4 --query count" "--config c4 --query location" "--config c4 --query count --opt leaf_vote=32" "--config c2 --query count"; do
  python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['opts'], r['ms_mean'])"
  done
done
