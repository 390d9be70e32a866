#!/bin/bash
# One MI355X profiling session of bench.py (run through gpurun from the repo root):
#   tools/gpu_profile.sh <tag> [bench args for the profiled (short) runs]
# Writes under gpurun_out/<tag>_*: the bench line, a kernel trace (+ per-kernel stats, concurrency timeline), and separate
# --pmc passes (SQ instruction/occupancy counters, FETCH_SIZE, WRITE_SIZE; never combined with other trace domains).
set -u
TAG=${1:-prof}; shift || true
ARGS=${*:---no-cpu --no-check --no-legs --steps 1 --warmup 1 --reps 2}
SUBFRAMES=${LSN_PROFILE_SUBFRAMES:-25600}   # subframes the profiled command processes: (steps + warmup) * nsf * reps
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
run_prof() {  # name, rocprofv3 options...
  local name=$1; shift
  rm -rf /tmp/lsnprof_$name
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d /tmp/lsnprof_$name -- python $R/bench.py $ARGS ) > $OUT/${TAG}_${name}.log 2>&1
  find /tmp/lsnprof_$name -name '*_results.db' | head -1
}
KT=$(run_prof kt --kernel-trace)
if [ -n "$KT" ]; then
  python tools/rocpd_summary.py $KT --tail 0.6 > $OUT/${TAG}_kernel_trace_stats.txt 2>&1
  python tools/timeline.py $KT --tail 0.6 > $OUT/${TAG}_timeline.txt 2>&1
fi
P1=$(run_prof pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU)
P2=$(run_prof pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_IFETCH)
P3=$(run_prof pmc_fetch --kernel-trace --pmc FETCH_SIZE)
P4=$(run_prof pmc_write --kernel-trace --pmc WRITE_SIZE)
[ -n "$P1$P2" ] && python tools/pmc_generic_summary.py $OUT/${TAG}_pmc_sq.json $P1 $P2 --subframes $SUBFRAMES > $OUT/${TAG}_pmc_sq.txt 2>&1
[ -n "$P3" ] && [ -n "$P4" ] && python tools/pmc_summary.py $P3 $P4 $OUT/${TAG}_pmc_hbm.json > $OUT/${TAG}_pmc_hbm.txt 2>&1
ls -la $OUT | tail -20
