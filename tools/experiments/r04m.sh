# where does the rate fall off with the batch size (round 3 saw 127 k at B = 512 and 118 k at B = 768 / 1024)?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-ingest --cpu-sample 0 --steps 20 "$@" > $O/r04m_$tag.json 2>> $O/r04m.err && python - $O/r04m_$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); B=d['config']['pairs_per_gpu_per_step']
print(sys.argv[2], d['value'], 'pairs/s', round(d['ms_per_step']/B*512,4), 'ms per 512 pairs | under schedule per 512:', {k:round(v['ms_per_step']/B*512,3) for k,v in d['stages'].items() if v['ms_per_step']}, '| alone per 512:', {k:round(v['ms_per_step']/B*512,3) for k,v in d['stages_isolated'].items() if v['ms_per_step']})
PY
}
run b512 --batch 512
run b576 --batch 576
run b640 --batch 640
run b768 --batch 768
run b1024 --batch 1024
run b1024_p4 --batch 1024 --tune pyr3_chunks=4
run b768_p4 --batch 768 --tune pyr3_chunks=4
run b512_again --batch 512
