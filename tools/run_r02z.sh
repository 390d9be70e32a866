python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02z_tests.txt
python bench.py --reps 5 > gpurun_out/r02z_bench.json 2> gpurun_out/r02z.err
