export LSN_FILE_DEBUG=1
python tools/file_replay_bench.py > gpurun_out/r02zf_default.txt 2>&1
LSN_FILE_MMAP=0 python tools/file_replay_bench.py > gpurun_out/r02zf_pread.txt 2>&1
LSN_FILE_SLOTS=8 python tools/file_replay_bench.py > gpurun_out/r02zf_slots8.txt 2>&1
LSN_FILE_SLOTS=8 LSN_FILE_BLOCK=3200 python tools/file_replay_bench.py > gpurun_out/r02zf_slots8_blk3200.txt 2>&1
