# measurement only (round 6): how the resident rate reacts to k_viterbi's work - the decode of every candidate run twice (same results, gated)
cd ${GRAFT_REPO_ROOT:-.}
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result"
run() { python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['pcap_diff'], d['host']['cores_busy_in_timed_region'], {k: round(v,1) for k,v in d['detail']['kernel_ms_per_6400_subframes'].items() if k in ('k_viterbi','k_turbo<128>','k_ofdm')})"; }
run base; run base
cd ltesniffer_amd/csrc; touch kernels/stage_a.hip; make FLAGS="$F -DLSN_EXP_VITERBI_TWICE" 2>&1 | grep -E "error" ; cd ../..
run twice; run twice
