#!/bin/bash
for r in 4 8 16 32 64 128; do
  timeout -k 10 120 python bench.py --tune pyr_rows=$r --tune no_pyr3=1 --no-overlap --steps 20 --warmup 3 --cpu-sample 0 --no-ingest > gpurun_out/sw_$r.log 2>&1 || exit 1
  python - <<PY
import json
l=json.loads(open('gpurun_out/sw_$r.log').read().strip().splitlines()[-1])
print($r, l['stages']['pyr'], l['stages']['gray']['ms_per_step'], l['value'])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pyr -o pyr -- python3 /root/repo/bench.py --no-overlap --steps 10 --warmup 2 --cpu-sample 0 > /root/repo/gpurun_out/prof_pyr.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_pyr/**/*kernel_trace.csv',recursive=True)[0]
import collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'pyr_down' in r['Kernel_Name']:
        d[(r['Grid_Size_X'],r['Grid_Size_Y'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in d.items(): print(k, len(v), sum(v)/len(v)/1e3,'us')
PY
