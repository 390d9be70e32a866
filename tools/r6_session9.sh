#!/bin/bash
# round 6, session 9: harq_mode = 1 with the retransmissions combined and decoded in batches ahead of the commit walk (harqScout / harqRunBatch): parity tests, then the gated HARQ leg
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] harq parity"; timeout 900 python -m pytest tests -m gpu -q -x -k "harq" 2>&1 | tail -15 | cut -c1-400
echo "[$(el) s] HARQ leg"; LSN_BENCH_LEGS=cfg3_16_dB_harq_mode_1,cfg3_at_16_dB_snr timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu > gpurun_out/r06_harq_leg.json 2> gpurun_out/r06_harq_leg.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_harq_leg.json").read().strip().splitlines()[-1])
for k in ("cfg3_16_dB_harq_mode_1", "cfg3_at_16_dB_snr"):
    print(k, json.dumps(d["first_h2d_to_last_pdu"].get(k)))
P
tail -3 gpurun_out/r06_harq_leg.err | cut -c1-300
echo "[$(el) s] done"
