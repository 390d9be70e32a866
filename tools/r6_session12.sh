#!/bin/bash
# round 6, session 12: the tree with candidate pruning, batched HARQ and the history-register trace-back: GPU suite, smoke, the driver's bench command, the round's profile set
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] GPU suite"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -6 gpurun_out/r06_gpu_tests.txt | cut -c1-600
echo "[$(el) s] smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "[$(el) s] bench line"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_n1.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
print({k: round(v, 2) for k, v in d["detail"]["kernel_ms_per_6400_subframes"].items()})
print({k: v for k, v in d["detail"]["per_6400_subframes"].items() if k in ("nof_candidate_misses", "nof_ondemand_decodes", "ms_search", "ms_search_core", "ms_commit")}, d["host"])
P
tail -3 gpurun_out/r06_bench_n1.err | cut -c1-300
echo "[$(el) s] profile set"; bash tools/gpu_profile.sh r06 > gpurun_out/r06_profile.log 2>&1; tail -3 gpurun_out/r06_profile.log
head -22 gpurun_out/r06_kernel_trace_stats.txt
echo "[$(el) s] done"
