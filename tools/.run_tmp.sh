timeout 280 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pbch.py tests/test_gpu_sync.py -x -q -m gpu -k "exhaustive or cfg3_20mhz or pbch or mib or small_cell or all_zero or cfi_1_2 or sync or 6prb" 2>&1 | tail -3
bash tools/ab_old_new.sh "lib_prev lib" 2
