R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_jpeg.py tests/test_gpu_comm.py -q -x -m gpu > $O/r04k_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04k_tests.log)"
TESTS="tests/test_gpu_pipeline.py" bash tools/experiments/run_variants.sh 2>&1 | grep -v "tests rc"
python - <<'PY'
import json
for v in ('p3w1','p3w2','p3w4'):
    for i in (1,2):
        d=json.load(open(f'gpurun_out/var_{v}_{i}.json')); print(v,i,d['ms_per_step'],'pyr alone',d['stages_isolated']['pyr']['ms_per_step'],'pyr sched',d['stages']['pyr']['ms_per_step'],'gray sched',d['stages']['gray']['ms_per_step'],'eig sched',d['stages']['eig']['ms_per_step'])
PY
