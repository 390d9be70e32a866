S=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02zi_bench.json 2> gpurun_out/r02zi.err
E=$(date +%s.%N); echo "wall $(echo "$E - $S" | bc)" > gpurun_out/r02zi_wall.txt
