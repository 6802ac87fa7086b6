R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r04e_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04e_tests.log)"
for c in c1 c2 c4; do
  timeout -k 10 600 python bench.py --config $c --no-ingest --steps 20 > $O/r04e_bench_$c.json 2> $O/r04e_bench_$c.err || { echo "bench $c failed"; tail -5 $O/r04e_bench_$c.err; }
  python tools/show_bench.py $O/r04e_bench_$c.json
done
cd /tmp && export TMPDIR=/tmp
for c in c2 c4; do
rm -rf $O/prof_r04e_$c
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04e_$c -o p -- python3 $R/bench.py --config $c --steps 5 --warmup 2 --cpu-sample 0 --no-ingest --no-overlap --streams 1 > $O/prof_r04e_$c.log 2>&1 || exit 1
cp $(find $O/prof_r04e_$c -name '*kernel_stats.csv' | head -1) $O/r04e_kernel_stats_serial_$c.csv
head -14 $O/r04e_kernel_stats_serial_$c.csv | cut -d, -f1-6
done
