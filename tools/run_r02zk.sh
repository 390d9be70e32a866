python bench.py --no-cpu --no-check --steps 2 --warmup 1 > gpurun_out/r02zk_bench.json 2> gpurun_out/r02zk.err
