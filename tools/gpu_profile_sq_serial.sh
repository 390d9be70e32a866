#!/bin/bash
# The two SQ counter passes of tools/gpu_profile.sh on ONE hardware queue (GPU_MAX_HW_QUEUES=1): rocprofv3 serialises profiled dispatches per queue,
# and with the 16 queues the engine normally runs on the per-dispatch durations of a counter pass include the wait for the other queues' kernels -
# "every kernel alone" (roofline.gpu_saturation) is only that on one queue.  Instruction counts do not depend on it.   tools/gpu_profile_sq_serial.sh <tag>
set -u
TAG=${1:-prof}
STEPS=${LSN_PROFILE_STEPS:-20}; WARM=${LSN_PROFILE_WARMUP:-5}; STEP_SF=${LSN_PROFILE_STEP_SF:-20000}; BATCH=${LSN_PROFILE_BATCH:-400}
ARGS="--gpus 1 --steps $STEPS --warmup $WARM --step-sf $STEP_SF --batch $BATCH --no-cpu --no-legs ${LSN_PROFILE_EXTRA:-}"
ALL_SF=$(((STEPS + WARM) * STEP_SF))
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
export GPU_MAX_HW_QUEUES=1
run_prof() { local name=$1; shift; rm -rf /tmp/lsnprof_$name; ( cd /tmp && timeout 600 rocprofv3 "$@" -d /tmp/lsnprof_$name -- python $R/bench.py $ARGS ) > $OUT/${TAG}_${name}.log 2>&1; find /tmp/lsnprof_$name -name '*_results.db' | head -1; }
P1=$(run_prof pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU)
P2=$(run_prof pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_IFETCH)
[ -n "$P1$P2" ] && python tools/pmc_generic_summary.py $OUT/${TAG}_pmc_sq.json $P1 $P2 --subframes $ALL_SF > $OUT/${TAG}_pmc_sq.txt 2>&1
python tools/tree_hash.py > $OUT/${TAG}_tree_hash.txt
head -30 $OUT/${TAG}_pmc_sq.txt | cut -c1-120
