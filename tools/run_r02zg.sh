export LSN_FILE_DEBUG=1
python tools/file_replay_bench.py > gpurun_out/r02zg_default.txt 2>&1
LSN_FILE_MMAP=0 python tools/file_replay_bench.py > gpurun_out/r02zg_pread.txt 2>&1
python -m pytest tests/test_file_source.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r02zg_tests.txt
