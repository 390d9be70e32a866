#!/bin/bash
# A/B runs of library variants on one GPU box (tools/ab_build.sh builds them): tools/ab_old_new.sh "<variant> ..." [rounds] [bench args]
# variant = directory suffix under ltesniffer_amd/ ("lib" = the product build); prints value, pcap_diff, oracle gate, cold rate and kernel spans
VARS=${1:-"lib_old lib"}; ROUNDS=${2:-2}; shift; shift
for i in $(seq 1 $ROUNDS); do
for v in $VARS; do
  export LSN_LIB_PATH=$PWD/ltesniffer_amd/$v/libltesniffer_amd.so
  timeout 280 python bench.py --no-legs --no-cpu --steps 20 --warmup 5 "$@" > gpurun_out/ab_${v}_$i.json 2> gpurun_out/ab_${v}_$i.err || tail -3 gpurun_out/ab_${v}_$i.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ab_${v}_$i.json'));k=d['detail']['kernel_ms_per_6400_subframes']
print('%-8s'%'$v',d['value'],d['pcap_diff'],d['parity'].get('timed_equals_oracle'),d['parity']['timed_digest'],d['parity']['timed_records'],'cold',d['cold_state']['subframes_per_s'],'t128 %.1f t64 %.1f vit %.1f rm %.1f demod %.1f'%(k['k_turbo<128>'],k['k_turbo<64>'],k['k_viterbi'],k['k_rm'],k['k_pdsch_demod']),'its',d['detail']['per_6400_subframes']['nof_turbo_iterations'],'ondemand',d['detail']['ondemand_at_commit_per_6400'],d['detail'].get('table_hints_engine_total'))
PY
done; done
