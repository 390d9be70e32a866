python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r02x_tests.txt
python bench.py --no-cpu --no-check --no-legs --steps 3 --warmup 1 --reps 4 > gpurun_out/r02x_bench.json 2> gpurun_out/r02x.err
LSN_NO_CB_SKIP=1 python bench.py --no-cpu --no-check --no-legs --steps 3 --warmup 1 --reps 4 > gpurun_out/r02x_bench_noskip.json 2>> gpurun_out/r02x.err
