#!/bin/bash
# block splitting (option split): timeline + plain timings on the headline image
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TRIRO_HIP_LIBRARY=$REPO/trimesh-ray-optix_amd/lib_var/timeline/libtriro_hip.so
for A in "" "--opt split=7" "--opt split=6" "--opt split=5" "--opt split=6 --opt split_steal=16" "--opt split=6 --opt split_steal=32" \
         "--opt tile=2" "--opt tile=2 --opt split=6" "--opt tile=2 --opt split=5" "--opt tile=2 --opt split=4"; do
  python scripts/exp_timeline.py $A 2>&1 | tail -1
done
unset TRIRO_HIP_LIBRARY
for A in "" "--opt split=7" "--opt split=6" "--opt split=5" "--opt tile=2" "--opt tile=2 --opt split=6" "--opt tile=2 --opt split=5" "--opt tile=2 --opt split=4"; do
  python scripts/run_query.py --config c5i --query closest $A 2>&1 | tail -1
done
