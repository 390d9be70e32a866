#!/bin/bash
# session 30: the GPU suite three more times on a fresh box (flakiness check of the shipped tree), smoke, the default bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/s30_tests_$i.txt 2>&1; echo "run $i rc $?"; tail -1 gpurun_out/s30_tests_$i.txt
done
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/s30_default_bench.json 2> gpurun_out/s30_default_bench.err; echo "default bench rc $?"
python - <<'P'
import json
d = json.loads(open("gpurun_out/s30_default_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["steps"], d["warmup"], d["pcap_diff"], d["x_realtime_first_h2d_to_last_pdu"], d["roofline"]["frac"], d["roofline"]["profile"])
P
