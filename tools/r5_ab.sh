#!/bin/bash
# Round-5 A/B session on one MI355X (run through gpurun from the repo root): the resident headline (no legs, no CPU leg) under a list of
# environment variants, interleaved base / variant so that box-to-box differences cancel.  usage: tools/r5_ab.sh <tag> "<VAR=val ...>" ...
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; shift
OUT=gpurun_out/${TAG}_ab.txt; : > $OUT
ARGS="--gpus 1 --steps ${AB_STEPS:-10} --warmup ${AB_WARMUP:-2} --step-sf ${AB_STEP_SF:-20000} --no-legs --no-cpu"
one() {  # label, env string, extra args
  local line
  line=$(env $2 timeout 600 python bench.py $ARGS $3 2>gpurun_out/${TAG}_ab_err.txt | grep '^{"metric"' | tail -1)
  python - "$1" "$line" >> $OUT <<'PY'
import json, sys
lab, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    p = d["parity"]
    k = d["detail"]["kernel_ms_per_6400_subframes"]
    print("%-44s %9.0f sf/s  bad %s/%s  it/sf %.1f  cores %.2f  span ms/6400: turbo %.0f demod %.0f rm %.0f prep %.0f  inflight %.2f" % (
        lab, d["value"], p.get("oracle_blocks_mismatching"), p.get("oracle_blocks_compared"), d["detail"]["per_6400_subframes"]["nof_turbo_iterations"] / 6400.0,
        d["host"]["cores_busy_in_timed_region"], k["k_turbo<128>"], k["k_pdsch_demod"], k["k_rm"], k["k_pdsch_prep"], d["roofline"]["mean_launches_in_flight"]))
except Exception as ex:
    print("%-44s FAILED %s %s" % (lab, ex, line[:200]))
PY
  tail -1 $OUT
}
WL=""
case "$*" in *WORKLOAD=16*) WL="--workload cfg3_at_16_dB_snr";; esac
for v in "$@"; do
  one "base" "LSN_X=1" "$WL"
  extra=""
  case "$v" in *BATCH=*) b=${v##*BATCH=}; b=${b%% *}; extra="--batch $b";; esac
  one "$v" "$v" "$extra $WL"
done
one "base" "LSN_X=1" "$WL"
cat $OUT
