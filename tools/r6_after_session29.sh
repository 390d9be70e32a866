#!/bin/bash
# local step behind tools/r6_session29.sh: gpurun_out/r06_* -> profiles/ under the names profiles/current.json lists, generated DESIGN tables
cd "$(dirname $0)/.."
for t in r06 r06_16dB; do
  for f in kernel_trace.json kernel_trace_stats.txt timeline.txt pmc_hbm.json pmc_hbm.txt pmc_sq.json pmc_sq.txt; do
    [ -s gpurun_out/${t}_$f ] && cp gpurun_out/${t}_$f profiles/${t}_$f
  done
  [ -s gpurun_out/${t}_kt_bench.json ] && cp gpurun_out/${t}_kt_bench.json profiles/${t}_bench_under_rocprof.json
done
cp gpurun_out/r06_gpu_tests.txt profiles/r06_gpu_tests.txt
sed -i '/amdgpu.ids/d' profiles/r06_gpu_tests.txt
grep -h '^{"metric"' gpurun_out/r06_bench_n1.json | tail -1 > profiles/r06_bench_n1.json
python - <<'P'
import json
c = json.load(open("profiles/current.json"))
c["tree_hash"] = open("gpurun_out/r06_tree_hash.txt").read().strip()
json.dump(c, open("profiles/current.json", "w"), indent=1)
print("profiles/current.json ->", c["tree_hash"])
P
python tools/tree_hash.py
python tools/bench_show.py profiles/r06_bench_n1.json --update-design
python tools/roofline_statement.py r06 --all-sf 500000 --update-design
python tools/roofline_statement.py r06_16dB --all-sf 80000 --marker ROOFLINE_16DB --update-design
