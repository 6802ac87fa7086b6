R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/r04_final_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04_final_tests.log)"
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/r04_final_bench.json 2> $O/r04_final_bench.err; python tools/show_bench.py $O/r04_final_bench.json 2>/dev/null | head -1
