python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02zh_tests.txt
