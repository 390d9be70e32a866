set -u
OUT=gpurun_out; TAG=r03z; R=$(pwd); export TMPDIR=/tmp
ARGS="--gpus 1 --steps 20 --warmup 5 --step-sf 4000 --batch 400 --no-cpu --no-legs"
run_prof() { local name=$1; shift; rm -rf /tmp/lsnprof_$name; ( cd /tmp && timeout 600 rocprofv3 "$@" -d /tmp/lsnprof_$name -- python $R/bench.py $ARGS ) > $OUT/${TAG}_${name}.log 2>&1; find /tmp/lsnprof_$name -name '*_results.db' | head -1; }
KT=$(run_prof kt --kernel-trace)
python tools/kernel_trace_json.py $KT --last-ofdm 200 --subframes 80000 --out $OUT/${TAG}_kernel_trace.json > $OUT/${TAG}_kernel_trace_stats.txt 2>&1
python tools/timeline.py $KT --tail 0.75 > $OUT/${TAG}_timeline.txt 2>&1
P1=$(run_prof pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU)
python tools/pmc_generic_summary.py $OUT/${TAG}_pmc_sq1.json $P1 --subframes 100000 > $OUT/${TAG}_pmc_sq1.txt 2>&1
head -45 $OUT/${TAG}_timeline.txt
