#!/bin/bash
# session 9: raised wave priority (s_setprio) for every kernel except the decoder - libraries built with -DLSN_LIGHT_WAVE_PRIO=3 / 1 (tools/ab/) against this tree's
cd ${GRAFT_REPO_ROOT:-.}
P3="LSN_LIB_PATH=$PWD/tools/ab/libltesniffer_amd_prio3.so"; P1="LSN_LIB_PATH=$PWD/tools/ab/libltesniffer_amd_prio1.so"
bash tools/r5_ab.sh r05b_session9 "$P3" "$P1" "$P3" | tail -7
AB_STEPS=3 AB_WARMUP=1 bash tools/r5_ab.sh r05b_session9_16dB "$P3 WORKLOAD=16" | tail -3
