python -m pytest tests/test_gpu_ul.py -x -q 2>&1 | tail -6 > gpurun_out/r02zo_tests.txt
python tools/ulmode_bench.py > gpurun_out/r02zo_ulbench.txt 2>&1
