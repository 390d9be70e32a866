#!/bin/bash
# which switch makes a flaky GPU test flaky: runs the given tests N times under each environment and counts failures
cd ${GRAFT_REPO_ROOT:-.}
K=${1:-"full_load or four_crs_ports_small"}; N=${2:-6}
out=gpurun_out/flaky.txt; : > $out
for v in "LSN_X=1" "LSN_DECODE_THREADS=8" "GPU_MAX_HW_QUEUES=4" "LSN_KERNEL_TIMING_PERIOD=1" "LSN_NO_CB_SKIP=1" "LSN_NO_PRESIZE=1"; do
  f=0
  for i in $(seq 1 $N); do
    env $v timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > /tmp/flaky_one.txt 2>&1 || { f=$((f+1)); grep -h "^FAILED\|^E   " /tmp/flaky_one.txt | cut -c1-160 | head -3 >> $out; }
  done
  echo "$v : $f failures of $N" >> $out
done
cat $out
