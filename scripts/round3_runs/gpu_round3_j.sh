#!/bin/bash
export TRIRO_HIP_LIBRARY=$GRAFT_REPO_ROOT/trimesh-ray-optix_amd/lib_var/libtriro_hip.so
timeout 120 python scripts/debug_usteal.py 2>&1 | grep -v amdgpu | tail -5; echo "rc=${PIPESTATUS[0]}"
