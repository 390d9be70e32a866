python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/bench_show.py --no-cpu
