#!/bin/bash
# new entropy decoder: parity tests, random sweep, per-kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
timeout -k 10 400 python3 -m pytest tests/test_jpeg.py -x -q -m gpu > $O/j2_tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/j2_tests.txt
grep -q " passed" $O/j2_tests.txt && ! grep -q "failed" $O/j2_tests.txt || exit 1
timeout -k 10 300 python3 tools/stress_jpeg.py 400 7 > $O/j2_stress.txt 2>&1; echo "stress rc=$?"; tail -3 $O/j2_stress.txt
timeout -k 10 300 bash tools/experiments/jpeg_chunks.sh 256 256 > $O/j2_kstats.txt 2>&1; tail -12 $O/j2_kstats.txt
