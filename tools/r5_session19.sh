#!/bin/bash
# session 19: k_viterbi with the rotating-label add-compare-select (one lane exchange per step, DPP) - the candidate-table / PBCH / parity tests first, then the
# resident headline interleaved with the library of the plain recursion (ltesniffer_amd/lib_bperm: -DLSN_VITERBI_BPERMUTE), every run gated
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pbch.py tests/test_gpu_stream_golden.py -m gpu -q -x > gpurun_out/r05d_viterbi_tests.txt 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r05d_viterbi_tests.txt | cut -c1-220
L=$PWD/ltesniffer_amd/lib_bperm/libltesniffer_amd.so
EXP_STEPS=3 EXP_WARMUP=2 bash tools/r5_exp.sh r05d_session19 'rotating labels||' "plain recursion|LSN_LIB_PATH=$L|" 'rotating labels||' "plain recursion|LSN_LIB_PATH=$L|" 'rotating labels||' "plain recursion|LSN_LIB_PATH=$L|" 'rotating labels 16 dB||--workload cfg3_at_16_dB_snr' "plain recursion 16 dB|LSN_LIB_PATH=$L|--workload cfg3_at_16_dB_snr" | cut -c1-200
