#!/bin/bash
# rocprofv3 kernel trace + PMC passes of ONE query of one config (scripts/run_query.py).
# usage (GPU box): scripts/profile_query.sh <tag> <run_query args...>   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 4 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/scripts/run_query.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $REPO/scripts/run_query.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/scripts/run_query.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $REPO/scripts/run_query.py $ARGS > $OUT/pmc_l2.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta -- python3 $REPO/scripts/run_query.py $ARGS > $OUT/pmc_ta.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/scripts/run_query.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/scripts/run_query.py $ARGS > $OUT/pmc_write.log 2>&1
cd $REPO && PROFILE_CMD="scripts/run_query.py $ARGS" python3 scripts/summarize_profile.py $TAG ${KERNEL_KEY:-k_query_direct} > $OUT/summary.txt 2>&1
tail -25 $OUT/summary.txt
