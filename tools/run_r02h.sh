python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r02h_tests.txt
tools/ubench/valu_peak > gpurun_out/r02h_valu_peak.txt 2>&1
