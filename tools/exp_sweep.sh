#!/bin/bash
# experiment: decode threads x chunk size on the resident headline (one box, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
run() {
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-legs --no-cpu --batch $B 2> gpurun_out/exp.err | tail -1 > gpurun_out/exp.json
  python - <<'P'
import json,os
d=json.loads(open("gpurun_out/exp.json").read())
print("threads", os.environ.get("LSN_DECODE_THREADS"), "batch", d["config"].get("chunk_subframes", "?"), d["value"], d["pcap_diff"], d["host"]["cores_busy_in_timed_region"], d["host"]["busiest_threads"].get("lsn-search"))
P
}
for rep in 1 2; do
for B in 400 500; do
for T in 10 12 14; do export LSN_DECODE_THREADS=$T; run; done
done
done
