python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r02w_tests.txt
python bench.py --no-cpu --no-check --no-legs --steps 3 --warmup 1 --reps 4 > gpurun_out/r02w_bench.json 2> gpurun_out/r02w.err
