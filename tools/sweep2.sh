for nd in 6 8; do TAG="ndec=$nd" LSN_DECODE_THREADS=$nd python tools/bench_show.py --no-cpu; done
