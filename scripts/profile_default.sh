#!/bin/bash
# kernel trace of the DEFAULT bench command (python3 bench.py: 1000 timed steps, 50 warm-up), so that the
# rocprofv3 average and bench.py's own HIP-event average come from the very same command line.
# usage: scripts/profile_default.sh <tag>
set -u
TAG=${1:-r01}; REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py > $OUT/trace.log 2>&1
tail -c 400 $OUT/trace.log
