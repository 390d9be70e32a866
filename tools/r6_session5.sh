#!/bin/bash
# round 6, session 5: host cost experiment - nap lengths of the polling waits and number of decode threads against throughput and busy cores (resident leg only)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
run() { # name, env...
  name=$1; shift
  echo "[$(el) s] $name"
  env "$@" LSN_BENCH_ALL_THREADS=1 timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --no-legs --no-cpu > gpurun_out/r06_host_$name.json 2> gpurun_out/r06_host_$name.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r06_host_$name.json").read().strip().splitlines()[-1])
print("$name", "value", d["value"], "pcap_diff", d.get("pcap_diff"), "cores", d["host"]["cores_busy_in_timed_region"])
P
  grep "\[threads\]" gpurun_out/r06_host_$name.err | cut -c1-600
}
run base LSN_QUIET=1
run nap200 LSN_DEC_NAP_US=200
run nap500 LSN_DEC_NAP_US=500
run nap200_front60 LSN_DEC_NAP_US=200 LSN_FRONT_NAP_US=60
run nap200_dec6 LSN_DEC_NAP_US=200 LSN_DECODE_THREADS=6
run nap500_dec6_front60 LSN_DEC_NAP_US=500 LSN_DECODE_THREADS=6 LSN_FRONT_NAP_US=60
run nap500_dec4_front60 LSN_DEC_NAP_US=500 LSN_DECODE_THREADS=4 LSN_FRONT_NAP_US=60
echo "[$(el) s] done"
