python bench.py > gpurun_out/bench_r01d.json 2> gpurun_out/bench_r01d.err; tail -2 gpurun_out/bench_r01d.err; cut -c1-400 gpurun_out/bench_r01d.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof4 -o r04 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof4.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof4.log | cut -c1-200
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write | head
