#!/bin/bash
# session 16: the tree with stage A at the decode chains' priority - GPU suite, variants (stage-A chunks in flight, decode threads, old priority), then the whole bench line (short) for the legs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05c_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r05c_gpu_tests.txt | cut -c1-200
bash tools/r5_exp.sh r05b_session16 'base||' 'stage A highest (rounds 1-4)|LSN_STAGE_A_PRIO=2|' 'eight stage-A chunks in flight|LSN_LIB_PATH=tools/ab/libltesniffer_amd_nsa8.so|' '10 decode threads|LSN_DECODE_THREADS=10|' '14 decode threads|LSN_DECODE_THREADS=14|' 'eight in flight, decode normal prio n/a||' 'base||' | tail -8
timeout 1200 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r05c_bench_short.json 2> gpurun_out/r05c_bench_short.err; echo "bench rc $?"
python tools/bench_show.py gpurun_out/r05c_bench_short.json | cut -c1-250 | tail -22
