python -m pytest tests/test_gpu_parity.py tests/test_gpu_ul.py tests/test_gpu_wide75.py -x -q 2>&1 | tail -4 > gpurun_out/r02zm_tests.txt
python bench.py --no-cpu --no-check --no-legs --steps 3 --warmup 1 > gpurun_out/r02zm_mixed.json 2> gpurun_out/r02zm.err
