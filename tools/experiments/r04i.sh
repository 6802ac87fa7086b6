R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
build_variants/sqrt_ex > $O/r04_sqrt_exhaustive.txt 2>&1; tail -2 $O/r04_sqrt_exhaustive.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/r04i_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04i_tests.log)"
for i in 1 2; do python bench.py --no-ingest --cpu-sample 0 --steps 40 > $O/r04i_bench_$i.json 2>> $O/r04i.err; python tools/show_bench.py $O/r04i_bench_$i.json | head -1; done
