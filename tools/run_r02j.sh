python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02j_tests.txt
for cfg in "200 6" "400 8" "800 8" "1600 8" "800 10"; do set -- $cfg
  LSN_DECODE_THREADS=$2 LSN_TRACE=gpurun_out/r02j_trace_$1_$2.txt timeout 300 python bench.py --no-cpu --no-check --steps 3 --warmup 1 --reps 4 --batch $1 > gpurun_out/r02j_bench_$1_$2.json 2> gpurun_out/r02j_bench.err
  python tools/trace_gantt.py gpurun_out/r02j_trace_$1_$2.txt --skip-ms 150 > gpurun_out/r02j_gantt_$1_$2.txt 2>&1
  rm -f gpurun_out/r02j_trace_$1_$2.txt
done
