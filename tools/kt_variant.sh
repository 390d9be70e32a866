#!/bin/bash
# kernel trace (rocprofv3 --kernel-trace, no counters) of the driver's bench command for library variants: tools/kt_variant.sh "<variant> ..." [bench args]
#   -> gpurun_out/kt_<variant>_{kernel_trace.json,kernel_trace_stats.txt,timeline.txt,bench.json}
VARS=${1:-"lib"}; shift
ARGS=${*:---gpus 1 --steps 20 --warmup 5 --step-sf 4000 --batch 400 --no-cpu --no-legs}
export TMPDIR=/tmp
R=$(pwd)
for v in $VARS; do
  export LSN_LIB_PATH=$R/ltesniffer_amd/$v/libltesniffer_amd.so
  rm -rf /tmp/lsnkt_$v
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/lsnkt_$v -- python $R/bench.py $ARGS ) > gpurun_out/kt_${v}.log 2>&1
  DB=$(find /tmp/lsnkt_$v -name '*_results.db' | head -1)
  grep -h '^{"metric"' gpurun_out/kt_${v}.log | tail -1 > gpurun_out/kt_${v}_bench.json
  python tools/kernel_trace_json.py $DB --last-ofdm 200 --subframes 80000 --out gpurun_out/kt_${v}_kernel_trace.json > gpurun_out/kt_${v}_kernel_trace_stats.txt 2>&1
  python tools/timeline.py $DB --tail 0.75 > gpurun_out/kt_${v}_timeline.txt 2>&1
  python -c "import json; d=json.load(open('gpurun_out/kt_${v}_bench.json')); print('$v', d['value'], d['pcap_diff'])"
done
