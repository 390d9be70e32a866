#!/bin/bash
# session 17 (re-entry of round 5): the summaries profiles/current.json lists, taken again on the FINAL tree (tools/tree_hash.py af9f6482...: stage A at the
# decode chains' priority) - the container of sessions 12-16 was replaced and their gpurun_out/ with it.  One box, one tree, stages in order of importance, each
# started only while the session's own clock allows (the round's GPU budget is nearly spent): kernel trace + HBM traffic of the driver's command, the GPU suite,
# SQ counters on one hardware queue, the driver's command itself (bench line with every leg), then the same profile for the 16 dB workload.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s); LIMIT=${LSN_SESSION_LIMIT:-1150}
el() { echo $(( $(date +%s) - T0 )); }
ok() { [ $(( $(el) + $1 )) -lt $LIMIT ]; }   # ok <seconds the next stage is expected to need>
echo "[$(el) s] r05 kernel trace + HBM"; LSN_PROFILE_SKIP_SQ=1 bash tools/gpu_profile.sh r05 > gpurun_out/r05_profile.log 2>&1; tail -3 gpurun_out/r05_profile.log | cut -c1-160
echo "[$(el) s] GPU suite"; timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r05_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r05_gpu_tests.txt | cut -c1-200
if ok 240; then echo "[$(el) s] r05 SQ on one queue"; bash tools/gpu_profile_sq_serial.sh r05 > gpurun_out/r05_sq.log 2>&1; grep -E "k_turbo" gpurun_out/r05_pmc_sq.txt | head -3 | cut -c1-140; fi
bash tools/r5_copy_profiles.sh > /dev/null
if ok 240; then echo "[$(el) s] bench line"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err; echo "bench rc $?"
  python tools/bench_show.py gpurun_out/r05_bench_n1.json | head -30 | cut -c1-260; fi
export LSN_PROFILE_STEPS=3 LSN_PROFILE_WARMUP=1 LSN_PROFILE_EXTRA="--workload cfg3_at_16_dB_snr"
if ok 200; then echo "[$(el) s] 16 dB kernel trace + HBM"; LSN_PROFILE_SKIP_SQ=1 bash tools/gpu_profile.sh r05_16dB > gpurun_out/r05_16dB_profile.log 2>&1; tail -3 gpurun_out/r05_16dB_profile.log | cut -c1-160; fi
if ok 150; then echo "[$(el) s] 16 dB SQ"; bash tools/gpu_profile_sq_serial.sh r05_16dB > gpurun_out/r05_16dB_sq.log 2>&1; grep -E "k_turbo" gpurun_out/r05_16dB_pmc_sq.txt | head -3 | cut -c1-140; fi
echo "[$(el) s] done"
