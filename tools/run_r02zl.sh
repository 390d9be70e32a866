python bench.py --no-cpu --no-check --no-legs --steps 3 --warmup 1 > gpurun_out/r02zl_base.json 2> gpurun_out/r02zl.err
LSN_LIB_PATH=$PWD/ltesniffer_amd/lib_tb8/libltesniffer_amd.so python bench.py --no-cpu --no-check --no-legs --steps 3 --warmup 1 > gpurun_out/r02zl_tb8.json 2>> gpurun_out/r02zl.err
