REPO=$(pwd)
for V in base q32; do
  if [ $V = base ]; then unset TRIRO_HIP_LIBRARY; else export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/$V/libtriro_hip.so; fi
  for A in "--config c3 --query any --steps 8" "--config c3 --query closest --steps 8" "--config c3 --query count --steps 8" "--config c5s --query closest --steps 8" "--config c5s --query any --steps 8" "--config c5i --query closest" "--config c5i --res 4096 --query closest --steps 8" "--config c2 --query closest" "--config c4 --query count"; do
  python scripts/run_query.py $A 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', r['config'], r['query'], r['rays'], r['ms_mean'], r['mrays_per_s'])"
  done
done
