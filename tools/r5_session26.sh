#!/bin/bash
# session 26: what one rank does with the host cores an N-rank run leaves it (bench.py narrows the pipeline when quota / ranks < 3.4 cores): the resident headline
# pinned to 2 / 3 / 4 / 6 / 8 cores with the decode-thread count bench.py's rule picks for that share.  Every run gated (8 x 20 000 subframes).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
: > gpurun_out/r05_cores_per_rank.txt
for c in 2 3 4 6 8; do
  [ $(( $(date +%s) - T0 )) -gt 150 ] && break
  if [ $c -lt 4 ]; then th=$(python -c "print(max(3, min(8, int(($c - 0.9) / 0.3))))"); else th=12; fi
  LSN_DECODE_THREADS=$th taskset -c 0-$((c-1)) timeout 100 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu --no-legs --gen-threads 16 2> /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('cores $c decode_threads $th value %.0f pcap_diff %s cores_busy %.2f' % (d['value'], d['pcap_diff'], d['host']['cores_busy_in_timed_region']))
" >> gpurun_out/r05_cores_per_rank.txt
done
cat gpurun_out/r05_cores_per_rank.txt
