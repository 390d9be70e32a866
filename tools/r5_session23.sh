#!/bin/bash
# session 23: instruction-cache counters of the decoder (k_turbo<128> is 74.8 KB of code; a CDNA instruction cache holds 64 KB and is shared by two CUs)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$(pwd)
( cd /tmp && timeout 60 rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_WAIT_IFETCH\|SQ_INST_LEVEL[A-Z_]*" | sort -u ) > gpurun_out/r05_icache_counters_avail.txt
cat gpurun_out/r05_icache_counters_avail.txt | tr '\n' ' '; echo
C=$(grep -E "^SQC_ICACHE_(REQ|HITS|MISSES|MISSES_DUPLICATE)$" gpurun_out/r05_icache_counters_avail.txt | tr '\n' ' ')
[ -z "$C" ] && { echo "no instruction-cache counters"; exit 0; }
export GPU_MAX_HW_QUEUES=1
rm -rf /tmp/lsnprof_ic
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/lsnprof_ic -- python $R/bench.py --gpus 1 --steps 3 --warmup 1 --step-sf 20000 --batch 400 --no-cpu --no-legs ) > gpurun_out/r05_icache.log 2>&1
DB=$(find /tmp/lsnprof_ic -name '*_results.db' | head -1)
[ -n "$DB" ] && python tools/pmc_generic_summary.py gpurun_out/r05_pmc_icache.json $DB --subframes 80000 > gpurun_out/r05_pmc_icache.txt 2>&1
grep -E "k_turbo|k_viterbi|k_pdsch_demod|k_rm|k_ofdm" gpurun_out/r05_pmc_icache.txt | cut -c1-130
