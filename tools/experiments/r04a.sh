set -x
cd $GRAFT_REPO_ROOT
TESTS="tests/test_gpu_image_parity.py tests/test_gpu_pipeline.py" bash tools/experiments/run_variants.sh > gpurun_out/r04a_variants.log 2>&1
# baseline schedule with the product lib (rd2 default build), alternating
for i in 1 2; do
python bench.py --cpu-sample 0 --no-ingest --steps 40 --tune no_bgr_eig=1 > gpurun_out/r04a_base_$i.json 2>> gpurun_out/r04a.err
python bench.py --cpu-sample 0 --no-ingest --steps 40 > gpurun_out/r04a_new_$i.json 2>> gpurun_out/r04a.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04a_*_?.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], {k:(v['ms_per_step']) for k,v in d['stages'].items()}, {k:(v['ms_per_step']) for k,v in d['stages_isolated'].items()})
PY
