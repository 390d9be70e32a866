#!/bin/bash
# session 7: the decoder with fewer vector instructions per trellis step (round 5, second half) - GPU suite, then interleaved runs of this tree's library
# against the library of tree f853cd1f (tools/ab/, built from the previous commit; LSN_LIB_PATH) at 30 dB and at 16 dB
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05b_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r05b_gpu_tests.txt | cut -c1-300
OLD="LSN_LIB_PATH=$PWD/tools/ab/libltesniffer_amd_f853cd1f.so"
bash tools/r5_ab.sh r05b_session7 "$OLD" "$OLD" | tail -5
AB_STEPS=3 AB_WARMUP=1 bash tools/r5_ab.sh r05b_session7_16dB "$OLD WORKLOAD=16" | tail -3
