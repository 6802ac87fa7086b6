R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_image_parity.py tests/test_gpu_pipeline.py -q -x > $O/r04n_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04n_tests.log)"
for i in 1 2 3; do
 for t in 0 1; do
  timeout -k 10 300 python bench.py --no-ingest --cpu-sample 0 --steps 40 --tune resp_stream=$t > $O/r04n_t${t}_$i.json 2>> $O/r04n.err
  python - $O/r04n_t${t}_$i.json $t <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('resp_stream', sys.argv[2], d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['stages'].items() if v['ms_per_step']})
PY
 done
done
for c in c2 c4; do for t in 0 1; do
  timeout -k 10 300 python bench.py --config $c --no-ingest --cpu-sample 0 --steps 20 --streams 1 --tune resp_stream=$t > $O/r04n_${c}_t${t}.json 2>> $O/r04n.err
  python - $O/r04n_${c}_t${t}.json "$c $t" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'])
PY
done; done
