#!/bin/bash
# gpurun_out/r05_* (tools/gpu_profile.sh, tools/gpu_profile_sq_serial.sh) -> profiles/ under the names profiles/current.json lists
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for t in r05 r05_16dB; do
  for f in kernel_trace.json kernel_trace_stats.txt timeline.txt pmc_hbm.json pmc_hbm.txt pmc_sq.json pmc_sq.txt; do
    [ -s gpurun_out/${t}_$f ] && cp gpurun_out/${t}_$f profiles/${t}_$f
  done
  [ -s gpurun_out/${t}_kt_bench.json ] && cp gpurun_out/${t}_kt_bench.json profiles/${t}_bench_under_rocprof.json
done
ls -la profiles | grep -c "r05_"
