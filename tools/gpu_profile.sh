#!/bin/bash
# One MI355X profiling session of bench.py (run through gpurun from the repo root):
#   tools/gpu_profile.sh <tag> [bench args]
# Default = the DRIVER's command (--gpus 1 --steps 20 --warmup 5) with the legs that run OUTSIDE its timed region switched off
# (--no-cpu --no-legs; the timed region is unchanged).  Writes under gpurun_out/<tag>_*: the bench line, a kernel trace reduced to the timed
# region (per-kernel avg + EXCLUSIVE time, JSON + text; concurrency timeline), and separate --pmc passes (SQ instruction / occupancy
# counters, FETCH_SIZE, WRITE_SIZE; never combined with other trace domains).  LSN_PROFILE_SKIP_SQ=1 leaves the two SQ passes out (kernel trace + HBM traffic only).
set -u
TAG=${1:-prof}; shift || true
STEPS=${LSN_PROFILE_STEPS:-20}; WARM=${LSN_PROFILE_WARMUP:-5}; STEP_SF=${LSN_PROFILE_STEP_SF:-20000}; BATCH=${LSN_PROFILE_BATCH:-400}
ARGS=${*:---gpus 1 --steps $STEPS --warmup $WARM --step-sf $STEP_SF --batch $BATCH --no-cpu --no-legs ${LSN_PROFILE_EXTRA:-}}   # LSN_PROFILE_EXTRA="--workload cfg3_at_16_dB_snr": the second operating point
TIMED_SF=$((STEPS * STEP_SF)); ALL_SF=$(((STEPS + WARM) * STEP_SF)); TIMED_CHUNKS=$((TIMED_SF / BATCH))
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
run_prof() {  # name, rocprofv3 options...
  local name=$1; shift
  rm -rf /tmp/lsnprof_$name
  ( cd /tmp && timeout 900 rocprofv3 "$@" -d /tmp/lsnprof_$name -- python $R/bench.py $ARGS ) > $OUT/${TAG}_${name}.log 2>&1
  find /tmp/lsnprof_$name -name '*_results.db' | head -1
}
KT=$(run_prof kt --kernel-trace)
if [ -n "$KT" ]; then
  grep -h '^{"metric"' $OUT/${TAG}_kt.log | tail -1 > $OUT/${TAG}_kt_bench.json
  python tools/kernel_trace_json.py $KT --last-ofdm $TIMED_CHUNKS --subframes $TIMED_SF --out $OUT/${TAG}_kernel_trace.json > $OUT/${TAG}_kernel_trace_stats.txt 2>&1
  python tools/timeline.py $KT --tail 0.75 > $OUT/${TAG}_timeline.txt 2>&1
fi
P1=""; P2=""
if [ -z "${LSN_PROFILE_SKIP_SQ:-}" ]; then  # LSN_PROFILE_SKIP_SQ=1: kernel trace + HBM traffic passes only
P1=$(run_prof pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU)
P2=$(run_prof pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_IFETCH)
fi
P3=$(run_prof pmc_fetch --kernel-trace --pmc FETCH_SIZE)
P4=$(run_prof pmc_write --kernel-trace --pmc WRITE_SIZE)
[ -n "$P1$P2" ] && python tools/pmc_generic_summary.py $OUT/${TAG}_pmc_sq.json $P1 $P2 --subframes $ALL_SF > $OUT/${TAG}_pmc_sq.txt 2>&1
[ -n "$P3" ] && [ -n "$P4" ] && python tools/pmc_summary.py $P3 $P4 $OUT/${TAG}_pmc_hbm.json > $OUT/${TAG}_pmc_hbm.txt 2>&1
# which tree these summaries describe: bench.py prints roofline figures taken from profiles/ only when this hash equals the running tree's
python tools/tree_hash.py > $OUT/${TAG}_tree_hash.txt
cat $OUT/${TAG}_tree_hash.txt
ls -la $OUT | grep ${TAG}_ | tail -20
