#!/bin/bash
# whole GPU suite + smoke + JPEG stress (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/suite_tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/suite_tests.txt
grep -q " passed" $O/suite_tests.txt && ! grep -q "failed\|error" $O/suite_tests.txt || exit 1
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python3 tools/stress_jpeg.py 3000 11 > $O/suite_stress_jpeg.txt 2>&1; tail -1 $O/suite_stress_jpeg.txt
