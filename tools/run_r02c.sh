python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02c_tests.txt
for b in 200 800; do
  LSN_TRACE=gpurun_out/r02c_trace_$b.txt timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 --batch $b > gpurun_out/r02c_bench_$b.json 2> gpurun_out/r02c_bench_$b.err
  python tools/trace_gantt.py gpurun_out/r02c_trace_$b.txt --skip-ms 150 > gpurun_out/r02c_gantt_$b.txt 2>&1
done
