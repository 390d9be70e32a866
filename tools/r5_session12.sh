#!/bin/bash
# the final session of round 5, one box, one tree: kernel trace + HBM traffic of the driver's command, SQ counters on one hardware queue, the same at 16 dB; then the
# summaries go where bench.py looks for them (profiles/, in the box's copy of the tree - the same copies are made at home from gpurun_out/) and the driver's command
# itself runs: its line is profiles/r05_bench_n1.json
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
LSN_PROFILE_SKIP_SQ=1 bash tools/gpu_profile.sh r05 > gpurun_out/r05_profile.log 2>&1; tail -3 gpurun_out/r05_profile.log
bash tools/gpu_profile_sq_serial.sh r05 > gpurun_out/r05_sq.log 2>&1; grep -E "k_turbo" gpurun_out/r05_pmc_sq.txt | head -3 | cut -c1-140
( export LSN_PROFILE_STEPS=3 LSN_PROFILE_WARMUP=1 LSN_PROFILE_EXTRA="--workload cfg3_at_16_dB_snr"
  LSN_PROFILE_SKIP_SQ=1 bash tools/gpu_profile.sh r05_16dB > gpurun_out/r05_16dB_profile.log 2>&1; tail -3 gpurun_out/r05_16dB_profile.log
  bash tools/gpu_profile_sq_serial.sh r05_16dB > gpurun_out/r05_16dB_sq.log 2>&1; grep -E "k_turbo" gpurun_out/r05_16dB_pmc_sq.txt | head -3 | cut -c1-140 )
bash tools/r5_copy_profiles.sh
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err; echo "bench rc $?"
python tools/bench_show.py gpurun_out/r05_bench_n1.json | head -30 | cut -c1-260
