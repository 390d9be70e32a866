#!/bin/bash
# session 20: the decoder at ONE wavefront per SIMD (259 registers, no spills: ltesniffer_amd/lib_w1, -DTB_WAVES_ATTR=) - two decoder workgroups per CU instead of four, but half of
# every SIMD's register file and of the CU's LDS stay free for the other kernels of the chains (at two per SIMD four decoder workgroups take ALL of a CU's registers)
cd ${GRAFT_REPO_ROOT:-.}
L=$PWD/ltesniffer_amd/lib_w1/libltesniffer_amd.so
EXP_STEPS=3 EXP_WARMUP=2 bash tools/r5_exp.sh r05d_session20 'base (two decoder wavefronts per SIMD)||' "one per SIMD (259 registers)|LSN_LIB_PATH=$L|" 'base||' "one per SIMD|LSN_LIB_PATH=$L|" 'base 16 dB||--workload cfg3_at_16_dB_snr' "one per SIMD 16 dB|LSN_LIB_PATH=$L|--workload cfg3_at_16_dB_snr" | cut -c1-200
