cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log | cut -c1-300
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log | cut -c1-300
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write
