#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 400 python3 -m pytest tests/test_jpeg.py -x -q -m gpu > $O/j3_tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/j3_tests.txt
grep -q " passed" $O/j3_tests.txt && ! grep -q "failed" $O/j3_tests.txt || exit 1
timeout -k 10 300 python3 tools/experiments/probe_ingest.py --batch 512 --reps 10 > $O/j3_probe.txt 2>&1; tail -12 $O/j3_probe.txt
timeout -k 10 300 python3 bench.py --cpu-sample 0 --steps 20 > $O/j3_bench.json 2> $O/j3_bench.err; python3 -c "
import json; d=json.load(open('$O/j3_bench.json')); print(d['value'], d['ms_per_step']); i=d['ingest_inclusive']; print(i['jpeg_decode_on_device']['value'], i['jpeg_double_buffered']['value'])"
