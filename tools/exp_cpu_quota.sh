#!/bin/bash
# experiment: the resident headline under a CPU quota (cgroup v2 cpu.max where the box lets us write it, else taskset as a fallback - which does not bind the HIP runtime's own threads)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
CG=/sys/fs/cgroup/lsnq
mount | grep -m1 cgroup2
mkdir -p $CG 2>/dev/null && echo "+cpu" > /sys/fs/cgroup/cgroup.subtree_control 2>/dev/null
ls $CG/cpu.max 2>/dev/null && HAVE=1 || HAVE=0
echo "cgroup cpu.max writable: $HAVE"
run() {  # cores threads
  local cores=$1 thr=$2
  export LSN_DECODE_THREADS=$thr
  if [ $HAVE = 1 ]; then
    echo "$((cores * 100000)) 100000" > $CG/cpu.max
    ( echo $BASHPID > $CG/cgroup.procs; exec timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-legs --no-cpu 2> gpurun_out/exp.err | tail -1 > gpurun_out/exp.json )
  else
    timeout 600 taskset -c 0-$((cores - 1)) python bench.py --gpus 1 --steps 10 --warmup 3 --no-legs --no-cpu 2> gpurun_out/exp.err | tail -1 > gpurun_out/exp.json
  fi
  python - <<P
import json
d=json.loads(open("gpurun_out/exp.json").read())
print("quota $cores cores, $thr decode threads:", d["value"], "subframes/s, pcap_diff", d["pcap_diff"], ", busy cores", d["host"]["cores_busy_in_timed_region"], ", search", d["host"]["busiest_threads"].get("lsn-search"))
P
}
run 16 12
run 4 12
run 4 8
run 3 8
run 3 6
run 2 6
run 2 4
[ $HAVE = 1 ] && cat $CG/cpu.stat | head -6
