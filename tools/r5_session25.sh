#!/bin/bash
# session 25 (after the final line): chunk size at 20 000-subframe steps - the resident headline with --batch 400 (default) / 600 / 800 / 1000, interleaved twice,
# every run gated on the cached oracle stream (13 x 20 000 subframes).  For the record only: the default stays what the profiles describe.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T0=$(date +%s)
: > gpurun_out/r05_batch_at_20000.txt
for rep in 1 2; do
  for b in 400 800 600 1000; do
    [ $(( $(date +%s) - T0 )) -gt 215 ] && break
    timeout 120 python bench.py --gpus 1 --steps 10 --warmup 3 --batch $b --no-cpu --no-legs 2> /dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('batch $b rep $rep value %.0f ms_per_step %.2f pcap_diff %s cores %.2f' % (d['value'], d['ms_per_step'], d['pcap_diff'], d['host']['cores_busy_in_timed_region']))
" >> gpurun_out/r05_batch_at_20000.txt
  done
done
cat gpurun_out/r05_batch_at_20000.txt
