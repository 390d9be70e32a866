python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02g_tests.txt
tools/ubench/valu_peak > gpurun_out/r02g_valu_peak.txt 2>&1
LSN_TRACE=gpurun_out/r02g_trace.txt timeout 600 python bench.py --steps 4 --warmup 1 --reps 4 > gpurun_out/r02g_bench.json 2> gpurun_out/r02g_bench.err
python tools/trace_gantt.py gpurun_out/r02g_trace.txt --skip-ms 150 > gpurun_out/r02g_gantt.txt 2>&1
rm -f gpurun_out/r02g_trace.txt
