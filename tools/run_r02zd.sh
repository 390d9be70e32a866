python -m pytest tests/test_gpu_ul.py tests/test_gpu_api.py tests/test_gpu_prach.py -x -q 2>&1 | tail -8 > gpurun_out/r02zd_tests.txt
python bench.py > gpurun_out/r02zd_bench.json 2> gpurun_out/r02zd.err
