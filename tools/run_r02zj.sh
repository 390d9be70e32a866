python -m pytest tests/test_gpu_ul.py tests/test_gpu_api.py -x -q 2>&1 | tail -8 > gpurun_out/r02zj_tests.txt
