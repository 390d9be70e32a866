python bench.py > gpurun_out/bench_final2.json 2> gpurun_out/bench_final2.err; tail -2 gpurun_out/bench_final2.err; cut -c1-300 gpurun_out/bench_final2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -o r05 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof5.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof5.log | cut -c1-200
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_valu2 -o v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_valu2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch2 -o f -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write2 -o w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_write2.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof5 $GRAFT_REPO_ROOT/gpurun_out/pmc_valu2 $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch2 $GRAFT_REPO_ROOT/gpurun_out/pmc_write2
