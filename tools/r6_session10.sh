#!/bin/bash
# round 6, session 10: where the commit turn of the batched HARQ path spends its time (the chunk-size sweep of profiles/r06_harq_batches.txt ran with a bench switch that is gone again)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for b in 400; do
LSN_BENCH_LEGS=cfg3_16_dB_harq_mode_1 timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu > gpurun_out/r06_harq_leg_$b.json 2> gpurun_out/r06_harq_leg_$b.err; echo "bench rc $?"
python - $b <<'P'
import json, sys
d=json.loads(open("gpurun_out/r06_harq_leg_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
v=d["first_h2d_to_last_pdu"].get("cfg3_16_dB_harq_mode_1")
print(sys.argv[1], {k: v.get(k) for k in ("subframes_per_s", "timed_s", "pcap_diff", "harq_combines", "harq_commit_ms", "error")})
P
done
