for i in 1 2; do
python bench.py --no-cpu --no-check --no-legs --steps 6 --warmup 1 > gpurun_out/r02zn_mixed$i.json 2>> gpurun_out/r02zn.err
LSN_LIB_PATH=$PWD/ltesniffer_amd/lib_tb8/libltesniffer_amd.so python bench.py --no-cpu --no-check --no-legs --steps 6 --warmup 1 > gpurun_out/r02zn_base$i.json 2>> gpurun_out/r02zn.err
done
