cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1e -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT && python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-600
