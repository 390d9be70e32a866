#!/bin/bash
# round 6, session 29: final tree of the round (session 24 + integer sample formats of the file source and of the host-buffer path) - GPU suite, smoke, the driver's bench command, the profile set of the headline and of the second operating point (16 dB)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] GPU suite"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r06_gpu_tests.txt | cut -c1-600
echo "[$(el) s] smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "[$(el) s] profile set"; bash tools/gpu_profile.sh r06 > gpurun_out/r06_profile.log 2>&1; tail -2 gpurun_out/r06_profile.log
echo "[$(el) s] profile set 16 dB"; LSN_PROFILE_STEPS=3 LSN_PROFILE_WARMUP=1 LSN_PROFILE_EXTRA='--workload cfg3_at_16_dB_snr' bash tools/gpu_profile.sh r06_16dB > gpurun_out/r06_16dB_profile.log 2>&1; tail -2 gpurun_out/r06_16dB_profile.log
echo "[$(el) s] bench line (profiles of this tree in place: current.json is rewritten first)"
python - <<'P'
import json
c = json.load(open("profiles/current.json"))
c["tree_hash"] = open("gpurun_out/r06_tree_hash.txt").read().strip()
json.dump(c, open("profiles/current.json", "w"), indent=1)
P
for f in kernel_trace.json kernel_trace_stats.txt timeline.txt pmc_hbm.json pmc_hbm.txt pmc_sq.json pmc_sq.txt; do cp gpurun_out/r06_$f profiles/r06_$f; done
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_n1.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
print(json.dumps(d["roofline"])[:1500])
P
tail -2 gpurun_out/r06_bench_n1.err | cut -c1-300
head -12 gpurun_out/r06_16dB_kernel_trace_stats.txt
echo "[$(el) s] done"
