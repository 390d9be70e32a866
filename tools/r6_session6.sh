#!/bin/bash
# round 6, session 6: does one GPU carry two independent cells?  two bench processes at once on the same device (each its own engine, capture, oracle gate),
# long timed regions (60 steps) and wall-clock stamps so that the overlap is on record
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
run1() { LSN_QUIET=1 LSN_BENCH_STAMP=1 timeout 500 python bench.py --gpus 1 --steps 60 --warmup 5 --no-legs --no-cpu > gpurun_out/r06_two_$1.json 2> gpurun_out/r06_two_$1.err; }
show() { python -c "
import json,sys
for n in sys.argv[1:]:
    d=json.loads(open('gpurun_out/r06_two_%s.json'%n).read().strip().splitlines()[-1]); print(n, d['value'], 'pcap_diff', d['pcap_diff'], 'cores', d['host']['cores_busy_in_timed_region'], 'timed s', d['detail']['timed_region_s'])
" "$@"; grep -h stamp $(for n in "$@"; do echo gpurun_out/r06_two_$n.err; done); }
echo "alone:"; run1 alone; show alone
echo "two at once:"; run1 a & run1 b & wait; show a b
echo "two at once, 6 decode threads each:"; LSN_DECODE_THREADS=6 run1 c & LSN_DECODE_THREADS=6 run1 d & wait; show c d
