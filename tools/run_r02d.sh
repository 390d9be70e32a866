python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02d_tests.txt
for b in 200 800; do for t in 6 8 10; do
  LSN_DECODE_THREADS=$t LSN_TRACE=gpurun_out/r02d_trace_${b}_$t.txt timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 --batch $b > gpurun_out/r02d_bench_${b}_$t.json 2> gpurun_out/r02d_bench_${b}_$t.err
  python tools/trace_gantt.py gpurun_out/r02d_trace_${b}_$t.txt --skip-ms 150 > gpurun_out/r02d_gantt_${b}_$t.txt 2>&1
  rm -f gpurun_out/r02d_trace_${b}_$t.txt
done; done
