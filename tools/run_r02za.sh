python -m pytest tests/test_gpu_ul.py tests/test_gpu_parity.py tests/test_gpu_prach.py -x -q 2>&1 | tail -8 > gpurun_out/r02za_tests.txt
