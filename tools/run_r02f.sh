python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02f_tests.txt
for cfg in "200 6" "200 8" "200 10" "400 8" "800 8"; do set -- $cfg
  LSN_DECODE_THREADS=$2 LSN_TRACE=gpurun_out/r02f_trace.txt timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 --batch $1 > gpurun_out/r02f_bench_$1_$2.json 2> gpurun_out/r02f_bench.err
  python tools/trace_gantt.py gpurun_out/r02f_trace.txt --skip-ms 150 > gpurun_out/r02f_gantt_$1_$2.txt 2>&1
done
rm -f gpurun_out/r02f_trace.txt
