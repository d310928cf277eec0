#!/bin/bash
# Evidence for the 4-wide-node experiment (branch exp/bvh4, extracted to scratch_bvh4/ and built there):
# its own bench with the binary kernel of that commit (wide=0) and with the 4-wide kernel (wide=1), plus a
# kernel trace + SQ counters of the 4-wide run.  usage (GPU box): scripts/exp_bvh4.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
B=$REPO/scratch_bvh4
OUT=$REPO/gpurun_out/prof_r02_bvh4
mkdir -p $OUT
cd $B
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --opt wide=0 2>/dev/null | tee $OUT/bench_binary.json
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --opt wide=1 --stats 2>/dev/null | tee $OUT/bench_wide4.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B/bench.py --steps 20 --warmup 3 --no-cpu-baseline --opt wide=1 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $B/bench.py --steps 20 --warmup 3 --no-cpu-baseline --opt wide=1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $B/bench.py --steps 20 --warmup 3 --no-cpu-baseline --opt wide=1 > $OUT/pmc_sq2.log 2>&1
cd $REPO && PROFILE_CMD="exp/bvh4: bench.py --steps 20 --warmup 3 --opt wide=1" python3 scripts/summarize_profile.py r02_bvh4 k_query4 > /dev/null
cp profiles/r02_bvh4_summary.* $OUT/
