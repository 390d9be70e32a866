python bench.py > gpurun_out/bench_r01c.json 2> gpurun_out/bench_r01c.err; tail -2 gpurun_out/bench_r01c.err; cut -c1-900 gpurun_out/bench_r01c.json
