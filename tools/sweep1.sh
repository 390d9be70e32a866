python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; tail -2 gpurun_out/bench_r01b.err; cat gpurun_out/bench_r01b.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o r03 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof3.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof3.log | cut -c1-200
