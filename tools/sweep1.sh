TAG=prefetch3 python tools/bench_show.py --no-cpu
python tools/search_bench.py 400 | tail -3
