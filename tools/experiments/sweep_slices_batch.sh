cd $GRAFT_REPO_ROOT
for s in 1 2 3 4; do python3 bench.py --cpu-sample 0 --no-ingest --no-isolated --streams $s > gpurun_out/b_s$s.json 2> gpurun_out/b_s$s.err; python3 -c "
import json
d=json.load(open('gpurun_out/b_s$s.json'))
print('streams', $s, d['value'], d['ms_per_step'])
"; done
for b in 128 512; do python3 bench.py --cpu-sample 0 --no-ingest --no-isolated --batch $b > gpurun_out/b_b$b.json 2> gpurun_out/b_b$b.err; python3 -c "
import json
d=json.load(open('gpurun_out/b_b$b.json'))
print('batch', $b, d['value'], d['ms_per_step'])
"; done
