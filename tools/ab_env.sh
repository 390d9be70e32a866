#!/bin/bash
# interleaved A/B of one library under two environments on one GPU box: tools/ab_env.sh "<env A>" "<env B>" [rounds]   (e.g. "LSN_NO_DEFER=1" "LSN_X=0")
cd ${GRAFT_REPO_ROOT:-.}
A=$1; B=$2; N=${3:-3}
out=gpurun_out/ab_env.txt; : > $out
for i in $(seq 1 $N); do
  for v in "$A" "$B"; do
    r=$(env $v timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['detail']; k=p['decode_jobs_by_kind_per_6400']
print(d['value'], 'pcap_diff', d['pcap_diff'], 'iters/6400', round(sum(k['iterations'])), 'unused', round(sum(k['iterations_unused'])), 'jobs', [round(x) for x in k['jobs']], 'hints', p.get('table_hints_engine_total'), 'ondemand', p.get('ondemand_at_commit_per_6400'), 'ms', p.get('ms_ondemand_commit'))")
    echo "$v : $r" >> $out
  done
done
cat $out
