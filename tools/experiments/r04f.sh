# schedule sweep of the other BASELINE configurations: slices x batch (round 4)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
for spec in "c2 1 1024" "c2 2 1024" "c2 3 1024" "c2 2 2048" "c2 2 512" "c4 1 128" "c4 2 128" "c4 1 256" "c4 2 256" "c4 1 64"; do
  set -- $spec
  for rep in 1 2; do
  timeout -k 10 600 python bench.py --config $1 --streams $2 --batch $3 --no-ingest --cpu-sample 0 --no-isolated --steps 20 > $O/r04f_$1_s$2_b$3_$rep.json 2> $O/r04f.err || { echo "$spec failed"; tail -3 $O/r04f.err; }
  python - $O/r04f_$1_s$2_b$3_$rep.json "$spec" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', {k:v['ms_per_step'] for k,v in d['stages'].items() if v['ms_per_step']})
PY
  done
done
