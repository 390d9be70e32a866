bash tools/gpu_profile.sh r02_final > gpurun_out/r02_final_profile.log 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak tools/ubench/valu_peak.hip > gpurun_out/r02_final_valu_build.log 2>&1 && timeout 120 /tmp/valu_peak > gpurun_out/r02_final_valu_peak.txt 2>&1
