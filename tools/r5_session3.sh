#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests_c.txt 2>&1; echo "gpu tests rc $?" ; tail -15 gpurun_out/r05_gpu_tests_c.txt | cut -c1-300
tools/r5_ab.sh r05s3 "LSN_DECODE_THREADS=16 GPU_MAX_HW_QUEUES=32" "LSN_DECODE_THREADS=14 GPU_MAX_HW_QUEUES=32" "LSN_KERNEL_TIMING_PERIOD=0" "BATCH=500"
# the second operating point (16 dB): kernel trace + HBM traffic passes, then the two SQ passes on one hardware queue
export LSN_PROFILE_STEPS=3 LSN_PROFILE_WARMUP=1 LSN_PROFILE_EXTRA="--workload cfg3_at_16_dB_snr"
LSN_PROFILE_SKIP_SQ=1 bash tools/gpu_profile.sh r05_16dB > gpurun_out/r05_16dB_profile.log 2>&1; tail -5 gpurun_out/r05_16dB_profile.log
bash tools/gpu_profile_sq_serial.sh r05_16dB > gpurun_out/r05_16dB_sq.log 2>&1; tail -12 gpurun_out/r05_16dB_sq.log | cut -c1-160
