#!/bin/bash
# session 22: more chunks of stage A in flight (LSN_NSTREAM_A = 6 / 8 instead of 4): the search thread spends a third of its time waiting for stage A (ms_wait_front)
cd ${GRAFT_REPO_ROOT:-.}
L6=$PWD/ltesniffer_amd/lib_nsa6/libltesniffer_amd.so; L8=$PWD/ltesniffer_amd/lib_nsa8/libltesniffer_amd.so
EXP_STEPS=3 EXP_WARMUP=2 bash tools/r5_exp.sh r05d_session22 'base (4 stage-A streams)||' "6 streams|LSN_LIB_PATH=$L6|" "8 streams|LSN_LIB_PATH=$L8|" 'base||' "6 streams|LSN_LIB_PATH=$L6|" "8 streams|LSN_LIB_PATH=$L8|" 'base 16 dB||--workload cfg3_at_16_dB_snr' "8 streams 16 dB|LSN_LIB_PATH=$L8|--workload cfg3_at_16_dB_snr" | cut -c1-200
