python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02_last_tests.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_last_bench.json 2> gpurun_out/r02_last.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_last_smoke.txt 2>&1
