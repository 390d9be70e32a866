#!/bin/bash
# round 6, session 22: k_pdsch_demod and k_rm address arithmetic in 32 bits with 24-bit multiplies (no quarter-rate 64-bit products) - parity (stage C taps, uplink, HARQ), headline + 16 dB quick bench, VALU counters
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
el() { echo $(( $(date +%s) - T0 )); }
echo "[$(el) s] parity"; timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ul.py tests/test_gpu_wide75.py tests/test_gpu_zz_p_a_feedback.py -m gpu -q -x > gpurun_out/s22_tests.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/s22_tests.txt | cut -c1-400
echo "[$(el) s] bench"
for w in "" "--workload cfg3_at_16_dB_snr"; do
  timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-legs --no-cpu $w 2> gpurun_out/s22_bench.err | tail -1 > gpurun_out/s22_bench.json
  python - <<'P'
import json
d=json.loads(open("gpurun_out/s22_bench.json").read())
k=d["detail"]["kernel_ms_per_6400_subframes"]
print(d["value"], d["pcap_diff"], d["host"]["cores_busy_in_timed_region"], {x:round(k[x],1) for x in ("k_viterbi","k_pdsch_demod","k_turbo<128>","k_rm")})
P
done
echo "[$(el) s] counters"
export TMPDIR=/tmp; R=$(pwd)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d /tmp/s22pmc -- python $R/bench.py --gpus 1 --steps 4 --warmup 1 --no-legs --no-cpu ) > gpurun_out/s22_pmc.log 2>&1
DB=$(find /tmp/s22pmc -name '*_results.db' | head -1)
python tools/pmc_generic_summary.py gpurun_out/s22_pmc.json $DB --subframes 100000 > gpurun_out/s22_pmc.txt 2>&1
grep -E "SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU" gpurun_out/s22_pmc.txt | cut -c1-110
echo "[$(el) s] done"
