#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests_d.txt 2>&1; echo "gpu tests rc $?" ; tail -8 gpurun_out/r05_gpu_tests_d.txt | cut -c1-300
tools/r5_ab.sh r05s4 "LSN_TURBO_NO_PAIRS=1" "LSN_TURBO_NO_PAIRS=1"
AB_STEPS=3 AB_WARMUP=1 tools/r5_ab.sh r05s4_16dB "LSN_TURBO_NO_PAIRS=1 WORKLOAD=16"
# serial SQ pass of the headline (small: 6 steps) to see the decoder's stand-alone time and instruction counts with the new layout
LSN_PROFILE_STEPS=6 LSN_PROFILE_WARMUP=2 bash tools/gpu_profile_sq_serial.sh r05mid > gpurun_out/r05mid_sq.log 2>&1; grep -E "k_turbo|k_viterbi" gpurun_out/r05mid_pmc_sq.txt | cut -c1-140
