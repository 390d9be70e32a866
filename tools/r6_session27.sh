#!/bin/bash
# session 27: ramped blocks at both ends of a file replay (LSN_FILE_RAMP, default 100; 0 = off) - tests, then interleaved A/B on both sample formats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_file_source.py tests/test_gpu_pbch.py tests/test_gpu_sync.py -m gpu -x -q > gpurun_out/s27_tests.txt 2>&1
tail -3 gpurun_out/s27_tests.txt
O=gpurun_out/s27_file_ramp.txt
: > $O
for round in 1 2; do
  for fmt in sc16 cf32; do
    for ramp in 0 100 50; do
      echo "== $fmt, 20000 subframes, LSN_FILE_RAMP=$ramp (round $round)" >> $O
      LSN_FILE_RAMP=$ramp timeout 300 python tools/file_replay_bench.py 20000 800 $fmt 2>&1 | grep replay >> $O
    done
  done
done
cat $O
