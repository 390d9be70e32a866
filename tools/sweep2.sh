python tools/bench_show.py --no-cpu
python tools/bench_show.py --no-cpu --sync-steps
