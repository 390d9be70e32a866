python -m pytest tests/test_file_source.py tests/test_gpu_pbch.py tests/test_gpu_sync.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r02ze_tests.txt
python bench.py --no-cpu --no-check --steps 1 --warmup 1 --reps 2 > gpurun_out/r02ze_bench_mmap.json 2> gpurun_out/r02ze.err
LSN_FILE_MMAP=0 python bench.py --no-cpu --no-check --steps 1 --warmup 1 --reps 2 > gpurun_out/r02ze_bench_pread.json 2>> gpurun_out/r02ze.err
