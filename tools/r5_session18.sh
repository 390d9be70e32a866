#!/bin/bash
# session 18: why is the first replay of a file on a fresh engine half as fast as the second although its block buffers are reserved (44 k against 99 k subframes/s)?
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{ echo "== rounds 2-5a: blocks reserved, not touched (LSN_FILE_NO_PRETOUCH=1)"; LSN_FILE_NO_PRETOUCH=1 LSN_FILE_DEBUG=1 timeout 300 python tools/file_cold_probe.py 2>&1 | grep -v "^W2\|amdgpu.ids" | cut -c1-160 | head -60
  echo "== blocks touched and copied once when they are reserved"; LSN_FILE_DEBUG=1 timeout 300 python tools/file_cold_probe.py 2>&1 | grep -v "^W2\|amdgpu.ids" | cut -c1-160 | head -60
  echo "== again, without the debug lines"; timeout 300 python tools/file_cold_probe.py 2>&1 | grep "replay\|prepare"
  LSN_FILE_NO_PRETOUCH=1 timeout 300 python tools/file_cold_probe.py 2>&1 | grep "replay\|prepare"; } > gpurun_out/r05_file_cold_probe.txt 2>&1
cat gpurun_out/r05_file_cold_probe.txt
