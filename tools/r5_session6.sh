#!/bin/bash
# the profile session of the final tree: GPU suite, kernel trace + HBM traffic of the driver's command, SQ counters on one hardware queue, the same at 16 dB
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_gpu_tests.txt 2>&1; echo "gpu tests rc $?" ; tail -4 gpurun_out/r05_gpu_tests.txt | cut -c1-300
LSN_PROFILE_SKIP_SQ=1 bash tools/gpu_profile.sh r05 > gpurun_out/r05_profile.log 2>&1; tail -3 gpurun_out/r05_profile.log
bash tools/gpu_profile_sq_serial.sh r05 > gpurun_out/r05_sq.log 2>&1; grep -E "k_turbo" gpurun_out/r05_pmc_sq.txt | head -3 | cut -c1-140
export LSN_PROFILE_STEPS=3 LSN_PROFILE_WARMUP=1 LSN_PROFILE_EXTRA="--workload cfg3_at_16_dB_snr"
LSN_PROFILE_SKIP_SQ=1 bash tools/gpu_profile.sh r05_16dB > gpurun_out/r05_16dB_profile.log 2>&1; tail -3 gpurun_out/r05_16dB_profile.log
bash tools/gpu_profile_sq_serial.sh r05_16dB > gpurun_out/r05_16dB_sq.log 2>&1; grep -E "k_turbo" gpurun_out/r05_16dB_pmc_sq.txt | head -3 | cut -c1-140
head -3 gpurun_out/r05_kernel_trace_stats.txt; tail -1 gpurun_out/r05_kernel_trace_stats.txt
