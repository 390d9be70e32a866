#!/bin/bash
# session 28: integer samples through host buffers (lsn_phy_process_host_int) - file / host tests, then the bench line with both int16 legs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_file_source.py tests/test_stream_golden.py -m gpu -x -q > gpurun_out/s28_tests.txt 2>&1
tail -3 gpurun_out/s28_tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s28_bench.json 2> gpurun_out/s28_bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/s28_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["pcap_diff"], d["x_realtime_first_h2d_to_last_pdu"])
print(json.dumps(d["summary"]["first_h2d_to_last_pdu"]))
for k in ("host_pinned_sc16", "file_replay_sc16"):
    print(k, json.dumps(d["first_h2d_to_last_pdu"].get(k))[:900])
P
tail -2 gpurun_out/s28_bench.err | cut -c1-300
