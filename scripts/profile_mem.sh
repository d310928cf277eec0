#!/bin/bash
# Memory-pipeline PMC passes (TA/TCP/TD) of the headline bench.  usage: scripts/profile_mem.sh <tag> [bench args]
set -u
TAG=${1:-mem}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline $*"
i=0
for C in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
         "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_m$i -- python3 $REPO/bench.py $ARGS > $OUT/pmc_m$i.log 2>&1
done
ls $OUT | head -30
