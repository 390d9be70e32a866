python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r02u_tests.txt
timeout 900 python bench.py --steps 5 --warmup 1 --reps 5 > gpurun_out/r02u_bench_full.json 2> gpurun_out/r02u_bench.err
