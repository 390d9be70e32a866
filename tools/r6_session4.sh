#!/bin/bash
# round 6, session 4: CFO correction / tracking inside k_ofdm (lsn_phy_set_cfo_correction) next to the oracle fed by the same loop rule; whole GPU suite
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] cfo tests"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfo" > gpurun_out/r06_gpu_cfo.txt 2>&1; echo "cfo rc $?"; tail -15 gpurun_out/r06_gpu_cfo.txt | cut -c1-600
echo "[$(el) s] GPU suite"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -5 gpurun_out/r06_gpu_tests.txt | cut -c1-400
echo "[$(el) s] done"
