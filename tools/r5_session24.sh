#!/bin/bash
# session 24 (the last of round 5): the GPU suite and the driver's command on the shipped tree with the 500 000-subframe oracle cache in place (20 steps of 20 000 subframes, every block gated)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err; echo "bench rc $?"
python tools/bench_show.py gpurun_out/r05_bench_n1.json | head -24 | cut -c1-230
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r05_gpu_tests.txt 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/r05_gpu_tests.txt | cut -c1-200
