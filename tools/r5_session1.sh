#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests.txt 2>&1; echo "gpu tests rc $?" ; tail -3 gpurun_out/r05_gpu_tests.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_full_a.json 2> gpurun_out/r05_bench_full_a.err; echo "bench rc $?"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05_bench_full_a.json") if l.startswith("{")][-1])
    print("value", d["value"], "pcap_diff", d["pcap_diff"], "bad", d["parity"].get("oracle_blocks_mismatching"), "of", d["parity"].get("oracle_blocks_compared"))
    for k, v in d["first_h2d_to_last_pdu"].items():
        if isinstance(v, dict):
            print(k, {a: b for a, b in v.items() if a in ("subframes_per_s", "oracle_blocks_compared", "oracle_blocks_mismatching", "pcap_diff", "error", "golden_note", "turbo_iterations_per_subframe")} or list(v)[:4])
except Exception as ex:
    print("bench parse failed", ex)
PY
tools/r5_ab.sh r05s1 "LSN_TURBO_MIN_LDS=54000" "LSN_LIGHT_STREAM=1" "LSN_RISKY_FIRST=1" "BATCH=800" "LSN_DECODE_THREADS=16" "LSN_TURBO_MIN_LDS=54000 LSN_LIGHT_STREAM=1" "LSN_TURBO_MIN_LDS=81000"
