python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02zc_tests.txt
python bench.py --reps 5 --no-legs --no-cpu > gpurun_out/r02zc_bench.json 2> gpurun_out/r02zc.err
