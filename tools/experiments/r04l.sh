R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_jpeg.py -q -m gpu > $O/r04l_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04l_tests.log)"
(timeout -k 10 900 python tools/stress_parity.py 6000 97 > $O/r04_stress_parity.log 2>&1; echo "stress_parity rc=$? $(tail -1 $O/r04_stress_parity.log)") &
sleep 200; echo "progress: $(tail -1 $O/r04_stress_parity.log)"
wait
timeout -k 10 600 python tools/stress_jpeg.py 6000 23 > $O/r04_stress_jpeg.log 2>&1; echo "stress_jpeg rc=$? $(tail -1 $O/r04_stress_jpeg.log)"
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')"
