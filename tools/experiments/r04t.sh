R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for rep in 1 2 3; do for spec in "eig_rows=0" "eig_rows=540" "eig_rows=1080" "eig_rows=360"; do
  timeout -k 10 300 python bench.py --no-ingest --cpu-sample 0 --steps 40 --no-isolated --tune $spec > $O/r04t.json 2>> $O/r04t.err
  python - $O/r04t.json "$spec" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:16s} {d['value']:9.1f} pairs/s {d['ms_per_step']:.4f} ms/step  {({k:v['ms_per_step'] for k,v in d['stages'].items() if v['ms_per_step']})}")
PY
done; done
