#!/bin/bash
# per-launch durations of the pyramid kernel (one line per level) from a rocprofv3 kernel trace of bench.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_pyr -o pyr -- python3 /root/repo/bench.py --no-overlap --steps 10 --warmup 2 --cpu-sample 0 > /root/repo/gpurun_out/prof_pyr.log 2>&1
cd /root/repo && python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_pyr/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0]
    d[(n,r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()): print(k, len(v), round(sum(v)/len(v)/1e3,1),'us')
PY
