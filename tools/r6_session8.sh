#!/bin/bash
# round 6, session 8: k_viterbi with the history-register trace-back (3.5 scalar-unit instructions per step instead of 10 / 19) and one DOT wait per four steps:
# GPU suite, the driver's bench command (four-port leg on its regenerated oracle cache), then the round's profile set (kernel trace + PMC passes) of this tree
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] viterbi parity first"; timeout 600 python -m pytest tests -m gpu -q -x -k "exhaustive or mib or pbch or cfg1" 2>&1 | tail -3 | cut -c1-300
echo "[$(el) s] GPU suite"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r06_gpu_tests.txt | cut -c1-400
echo "[$(el) s] bench line"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_n1.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
print({k: round(v, 2) for k, v in d["detail"]["kernel_ms_per_6400_subframes"].items()})
P
tail -3 gpurun_out/r06_bench_n1.err | cut -c1-300
echo "[$(el) s] profile set"; bash tools/gpu_profile.sh r06 > gpurun_out/r06_profile.log 2>&1; tail -5 gpurun_out/r06_profile.log
head -40 gpurun_out/r06_kernel_trace_stats.txt
head -40 gpurun_out/r06_pmc_sq.txt
echo "[$(el) s] done"
