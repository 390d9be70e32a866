#!/bin/bash
# round 6, session 1: the tree after the collection pin (HARQ last-TBS in the commit), the removed A/B switches and the new bench line - GPU suite, then the driver's command
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] GPU suite"; timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r06_gpu_tests.txt | cut -c1-300
echo "[$(el) s] smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "[$(el) s] bench line"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc $?"
tail -c 1800 gpurun_out/r06_bench_n1.json; echo; tail -5 gpurun_out/r06_bench_n1.err | cut -c1-300
echo "[$(el) s] done"
