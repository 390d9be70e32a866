python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02zb_tests.txt
python bench.py --reps 5 --no-legs > gpurun_out/r02zb_bench.json 2> gpurun_out/r02zb.err
