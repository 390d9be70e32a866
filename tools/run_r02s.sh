python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r02s_tests.txt
for cfg in "400" "800"; do
  LSN_TRACE=gpurun_out/r02s_trace_$cfg.txt timeout 300 python bench.py --no-cpu --no-check --steps 3 --warmup 1 --reps 4 --batch $cfg > gpurun_out/r02s_bench_$cfg.json 2> gpurun_out/r02s_bench.err
  python tools/trace_gantt.py gpurun_out/r02s_trace_$cfg.txt --skip-ms 150 > gpurun_out/r02s_gantt_$cfg.txt 2>&1
  rm -f gpurun_out/r02s_trace_$cfg.txt
done
timeout 600 python bench.py --steps 5 --warmup 1 --reps 5 > gpurun_out/r02s_bench_full.json 2>> gpurun_out/r02s_bench.err
