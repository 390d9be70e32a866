#!/bin/bash
# session 25: the integer sample formats of the file source (tests) + the bench line with the file_replay_sc16 leg
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_file_source.py -m gpu -x -q > gpurun_out/s25_tests.txt 2>&1
tail -5 gpurun_out/s25_tests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s25_bench.json 2> gpurun_out/s25_bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/s25_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["pcap_diff"])
print(json.dumps(d["summary"]["first_h2d_to_last_pdu"]))
print(json.dumps(d["first_h2d_to_last_pdu"].get("file_replay_sc16")))
P
