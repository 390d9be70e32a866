#!/bin/bash
# Builds a variant of the product library for A/B measurements without touching ltesniffer_amd/lib:
#   tools/ab_build.sh <tag> [extra hipcc flags for kernels/stage_c.hip and kernels/stage_a.hip ...]
#   e.g. tools/ab_build.sh w1 '-DTB_WAVES_ATTR='   (the turbo kernels at one wavefront per SIMD); tools/ab_build.sh vit2 -DLSN_VITERBI_PAIRED   (k_viterbi with two candidates per wavefront)
# -> ltesniffer_amd/lib_<tag>/libltesniffer_amd.so (git-ignored, travels with gpurun); run with
#   LSN_LIB_PATH=$PWD/ltesniffer_amd/lib_<tag>/libltesniffer_amd.so python bench.py --no-cpu --no-legs      (or tools/ab_old_new.sh "lib lib_<tag>")
# every run is gated on the oracle (pcap_diff 0 = identical records on that workload).
set -e
cd "$(dirname "$0")/../ltesniffer_amd/csrc"
TAG=$1; shift
make -s
mkdir -p _build_$TAG ../lib_$TAG
for k in stage_c stage_a; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result "$@" -c kernels/$k.hip -o _build_$TAG/$k.o
done
OBJ=$(ls _build/*.o | grep -v -e stage_c.o -e stage_a.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$TAG/libltesniffer_amd.so $OBJ _build_$TAG/stage_c.o _build_$TAG/stage_a.o -lpthread
echo "built ltesniffer_amd/lib_$TAG/libltesniffer_amd.so"
