python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/bench_show.py --no-cpu
python tools/bench_show.py --no-cpu --nsf 12800 --steps 3
