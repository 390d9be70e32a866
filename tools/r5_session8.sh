#!/bin/bash
# session 8: what bounds the pipelined engine after the decoder lost 11 % of its instructions for 2-3 % of throughput?  Decoder clock ticks per subframe in
# the pipelined engine (instrumented library), two engines on ONE device, smaller chunks, fewer chains.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
OUT=gpurun_out/r05b_session8.txt; : > $OUT
ARGS="--gpus 1 --steps 10 --warmup 2 --step-sf 20000 --no-legs --no-cpu"
one() {  # label, env string, extra args
  local line
  line=$(env $2 timeout 600 python bench.py $ARGS $3 2>gpurun_out/r05b_session8_err.txt | grep '^{"metric"' | tail -1)
  echo "$line" > gpurun_out/r05b_session8_$(echo "$1" | tr -c 'A-Za-z0-9\n' '_').json
  python - "$1" "$line" >> $OUT <<'PY'
import json, sys
lab, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    p = d["parity"]
    k = d["detail"]["kernel_ms_per_6400_subframes"]
    print("%-40s %9.0f sf/s  bad %s/%s  it/sf %.1f  cores %.2f  span ms/6400: turbo %.0f demod %.0f rm %.0f prep %.0f  stage_c %.0f wait_slot %.0f  ticks %s" % (
        lab, d["value"], p.get("oracle_blocks_mismatching"), p.get("oracle_blocks_compared"), d["detail"]["per_6400_subframes"]["nof_turbo_iterations"] / 6400.0,
        d["host"]["cores_busy_in_timed_region"], k["k_turbo<128>"], k["k_pdsch_demod"], k["k_rm"], k["k_pdsch_prep"],
        d["detail"]["per_6400_subframes"]["ms_stage_c"], d["detail"]["per_6400_subframes"]["ms_wait_slot"], d["detail"].get("turbo_clock_ticks_per_subframe")))
except Exception as ex:
    print("%-40s FAILED %s %s" % (lab, ex, line[:200]))
PY
  tail -1 $OUT
}
CYC="LSN_LIB_PATH=$PWD/tools/ab/libltesniffer_amd_cycles.so"
one "base" "LSN_X=1" ""
one "instrumented library" "$CYC" ""
one "two engines on device 0" "LSN_BENCH_DEVICES=0,0" "--shard capture"
one "chunks of 200" "LSN_X=1" "--batch 200"
one "chunks of 300" "LSN_X=1" "--batch 300"
one "8 decode threads" "LSN_DECODE_THREADS=8" ""
one "10 decode threads" "LSN_DECODE_THREADS=10" ""
one "base" "LSN_X=1" ""
one "instrumented library 16 dB" "$CYC" "--workload cfg3_at_16_dB_snr --steps 3 --warmup 1"
one "two engines on device 0 16 dB" "LSN_BENCH_DEVICES=0,0" "--shard capture --workload cfg3_at_16_dB_snr --steps 3 --warmup 1"
