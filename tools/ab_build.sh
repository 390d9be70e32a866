#!/bin/bash
# Builds a variant of the product library for A/B measurements without touching ltesniffer_amd/lib:
#   tools/ab_build.sh <tag> [extra hipcc flags for kernels/stage_c.hip ...]
#   e.g. tools/ab_build.sh tb8 -DTB_S128=8 '-DTB_WAVES_ATTR=__attribute__((amdgpu_waves_per_eu(2,2)))'
# -> ltesniffer_amd/lib_<tag>/libltesniffer_amd.so (git-ignored, travels with gpurun); run with
#   LSN_LIB_PATH=$PWD/ltesniffer_amd/lib_<tag>/libltesniffer_amd.so python bench.py --no-cpu --no-check --no-legs
# bench.py prints the digest of the timed record stream (parity.timed_digest): equal digests = identical records on that workload.
set -e
cd "$(dirname "$0")/../ltesniffer_amd/csrc"
TAG=$1; shift
make -s
mkdir -p _build_$TAG ../lib_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result "$@" -c kernels/stage_c.hip -o _build_$TAG/stage_c.o
OBJ=$(ls _build/*.o | grep -v stage_c.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$TAG/libltesniffer_amd.so $OBJ _build_$TAG/stage_c.o -lpthread
echo "built ltesniffer_amd/lib_$TAG/libltesniffer_amd.so"
