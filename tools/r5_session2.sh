#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests_b.txt 2>&1; echo "gpu tests rc $?" ; tail -15 gpurun_out/r05_gpu_tests_b.txt | cut -c1-300
tools/r5_ab.sh r05s2 "LSN_TURBO_NO_PAIRS=1" "LSN_LIGHT_STREAM=1 GPU_MAX_HW_QUEUES=32" "GPU_MAX_HW_QUEUES=32" "LSN_TURBO_NO_PAIRS=1"
# the second operating point, 16 dB: pairs on / off
AB_STEPS=3 AB_WARMUP=1 tools/r5_ab.sh r05s2_16dB "LSN_TURBO_NO_PAIRS=1 WORKLOAD=16"
