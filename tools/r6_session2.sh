#!/bin/bash
# round 6, session 2: after the any-LCID connection-setup learning of the unknown-table branch (found by the decode pin) - GPU suite and the driver's command against the
# cached oracle streams (made before the change: a pcap_diff of 0 over 500 000 subframes + all legs says the change does not touch those streams)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] GPU suite"; timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r06_gpu_tests.txt | cut -c1-300
echo "[$(el) s] bench line"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc $?"
tail -c 1700 gpurun_out/r06_bench_n1.json; echo; tail -3 gpurun_out/r06_bench_n1.err | cut -c1-300
echo "[$(el) s] done"
