#!/bin/bash
# round 6, session 3: GPU suite with the multipath fading streams (TS 36.101 B.2 EPA / EVA / ETU through tools/txgen) next to the oracle
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] fading tests"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fading" > gpurun_out/r06_gpu_fading.txt 2>&1; echo "fading rc $?"; tail -5 gpurun_out/r06_gpu_fading.txt | cut -c1-400
echo "[$(el) s] GPU suite"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -5 gpurun_out/r06_gpu_tests.txt | cut -c1-400
echo "[$(el) s] done"
