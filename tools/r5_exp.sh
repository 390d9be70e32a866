#!/bin/bash
# Round-5 experiment runner on one MI355X (through gpurun from the repo root): the resident headline, gated, once per argument "label|ENV=val ...|extra bench args"
# usage: tools/r5_exp.sh <tag> "label|env|args" ...      -> gpurun_out/<tag>.txt (+ the JSON line of every run)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
TAG=$1; shift
OUT=gpurun_out/$TAG.txt; : > $OUT
ARGS="--gpus 1 --steps ${EXP_STEPS:-10} --warmup ${EXP_WARMUP:-2} --step-sf 20000 --no-legs --no-cpu"
for spec in "$@"; do
  IFS='|' read -r lab envs extra <<< "$spec"
  line=$(env ${envs:-LSN_X=1} timeout 600 python bench.py $ARGS $extra 2>gpurun_out/${TAG}_err.txt | grep '^{"metric"' | tail -1)
  echo "$line" > gpurun_out/${TAG}_$(echo "$lab" | tr -c 'A-Za-z0-9\n' '_').json
  python - "$lab" "$line" >> $OUT <<'PY'
import json, sys
lab, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    p = d["parity"]
    k = d["detail"]["kernel_ms_per_6400_subframes"]
    q = d["detail"]["per_6400_subframes"]
    print("%-34s %9.0f sf/s  bad %s/%s  it/sf %.1f  cores %.2f  span ms/6400: turbo %.0f demod %.0f rm %.0f prep %.0f | stage_c %.0f wait front %.0f slot %.0f commit %.0f  ticks %s" % (
        lab, d["value"], p.get("oracle_blocks_mismatching"), p.get("oracle_blocks_compared"), q["nof_turbo_iterations"] / 6400.0,
        d["host"]["cores_busy_in_timed_region"], k["k_turbo<128>"], k["k_pdsch_demod"], k["k_rm"], k["k_pdsch_prep"],
        q["ms_stage_c"], q["ms_wait_front"], q["ms_wait_slot"], q["ms_commit"], d["detail"].get("turbo_clock_ticks_per_subframe")))
except Exception as ex:
    print("%-34s FAILED %s %s" % (lab, ex, line[:200]))
PY
  tail -1 $OUT
done
