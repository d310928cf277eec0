#!/bin/bash
mkdir -p gpurun_out
python scripts/run_query.py --config c5s --query closest --each --steps 12 > gpurun_out/r3b_c5s.json 2>&1; cat gpurun_out/r3b_c5s.json
python scripts/run_query.py --config c3 --query closest --each --steps 12 > gpurun_out/r3b_c3.json 2>&1; cat gpurun_out/r3b_c3.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b_trace -- python3 $GRAFT_REPO_ROOT/scripts/run_query.py --config c5s --query closest --steps 12 > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r3b_trace/*/*_kernel_stats.csv | cut -c1-220 | head -20
