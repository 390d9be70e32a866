#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_sweep.txt; : > $out
run() { # label, env..., -- args
  local label=$1; shift
  local r=$(env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-legs $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['detail']
print(d['value'], d['pcap_diff'], p.get('turbo_iterations_per_subframe'), p.get('tb_decodes_per_subframe'))")
  echo "$label $r" >> $out
}
for b in 400 200 250 500 800 400; do ARGS="--batch $b" run "batch=$b threads=8" LSN_X=1; done
for t in 6 10 12; do ARGS="--batch 400" run "batch=400 threads=$t" LSN_DECODE_THREADS=$t; done
for t in 12; do ARGS="--batch 200" run "batch=200 threads=$t" LSN_DECODE_THREADS=$t; done
cat $out
