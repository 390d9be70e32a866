python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for n in 4 6; do TAG="ndec=$n b100" LSN_DECODE_THREADS=$n python tools/bench_show.py --steps 5 --warmup 2 --nsf 1600 --batch 100 --no-cpu; done
for n in 6; do TAG="ndec=$n b200" LSN_DECODE_THREADS=$n python tools/bench_show.py --steps 5 --warmup 2 --nsf 1600 --batch 200 --no-cpu; done
for n in 6; do TAG="ndec=$n b50" LSN_DECODE_THREADS=$n python tools/bench_show.py --steps 5 --warmup 2 --nsf 1600 --batch 50 --no-cpu; done
