#!/bin/bash
# round 6, session 7: GPU suite on the tree with the factored uplink trial plan / counting sort / incremental CRC; the driver's bench command with the new EVA70 leg
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] GPU suite"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r06_gpu_tests.txt | cut -c1-400
echo "[$(el) s] bench line"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_n1.json").read().strip().splitlines()[-1])
print("value", d["value"], "pcap_diff", d["pcap_diff"], "cores", d["host"]["cores_busy_in_timed_region"])
for k,v in d.get("other_configs", d.get("legs", {})).items() if isinstance(d.get("other_configs", d.get("legs", {})), dict) else []:
    print(k, {x: v.get(x) for x in ("subframes_per_s", "pcap_diff", "golden_note", "error") if x in v})
P
tail -c 1500 gpurun_out/r06_bench_n1.json; echo; tail -3 gpurun_out/r06_bench_n1.err | cut -c1-300
echo "[$(el) s] done"
