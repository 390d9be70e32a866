#!/bin/bash
# session 26: the 16-bit file source - block size / reader count / capture length (ungated rate probe, tools/file_replay_bench.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/s26_sc16_blocks.txt
: > $O
export GPU_MAX_HW_QUEUES=16
for cfg in "800 12" "400 12" "200 12" "400 6" "1600 12"; do
  set -- $cfg
  echo "== sc16, 20000 subframes, LSN_FILE_BLOCK=$1 LSN_FILE_READERS=$2" >> $O
  LSN_FILE_BLOCK=$1 LSN_FILE_READERS=$2 timeout 300 python tools/file_replay_bench.py 20000 800 sc16 >> $O 2>&1
done
echo "== sc16, 80000 subframes, LSN_FILE_BLOCK=800" >> $O
LSN_FILE_BLOCK=800 timeout 300 python tools/file_replay_bench.py 80000 800 sc16 >> $O 2>&1
echo "== sc16, 80000 subframes, LSN_FILE_BLOCK=400" >> $O
LSN_FILE_BLOCK=400 timeout 300 python tools/file_replay_bench.py 80000 800 sc16 >> $O 2>&1
echo "== cf32, 20000 subframes, LSN_FILE_BLOCK=800 (the shipped default)" >> $O
timeout 300 python tools/file_replay_bench.py 20000 800 cf32 >> $O 2>&1
grep -v amdgpu.ids $O
