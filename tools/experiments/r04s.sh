R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for spec in "eig_rows=0" "eig_rows=180" "eig_rows=216" "eig_rows=360" "eig_rows=540" "pyr3_chunks=2" "pyr3_chunks=8" "eig_rows=0"; do
  timeout -k 10 300 python bench.py --no-ingest --cpu-sample 0 --steps 40 --tune $spec > $O/r04s.json 2>> $O/r04s.err
  python - $O/r04s.json "$spec" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(f"{sys.argv[2]:16s} {d['value']:9.1f} pairs/s {d['ms_per_step']:.4f} ms/step  eig alone {d['stages_isolated']['eig']['ms_per_step']}")
PY
done
