for cfg in "200 6 0" "200 8 0" "200 6 1" "400 6 0" "400 8 0" "800 8 0"; do set -- $cfg
  LSN_TURBO_FORK=$3 LSN_DECODE_THREADS=$2 LSN_TRACE=gpurun_out/r02e_trace.txt timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 --batch $1 > gpurun_out/r02e_bench_$1_$2_$3.json 2> gpurun_out/r02e_bench.err
  python tools/trace_gantt.py gpurun_out/r02e_trace.txt --skip-ms 150 > gpurun_out/r02e_gantt_$1_$2_$3.txt 2>&1
done
rm -f gpurun_out/r02e_trace.txt
