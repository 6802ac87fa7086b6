# timeline + VALU count of the BGR-reading response kernel's schedule (round 4 experiment)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tl_bgr $O/tl_gray $O/pmc_bgr
rocprofv3 --kernel-trace --output-format csv -d $O/tl_bgr -o t -- python3 $R/bench.py --cpu-sample 0 --no-ingest --no-isolated --steps 10 > $O/tl_bgr.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/tl_gray -o t -- python3 $R/bench.py --cpu-sample 0 --no-ingest --no-isolated --steps 10 --tune no_bgr_eig=1 > $O/tl_gray.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_bgr -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-ingest --no-overlap --streams 1 > $O/pmc_bgr.log 2>&1 || exit 1
cd $R
python3 tools/timeline.py $(find $O/tl_bgr -name '*kernel_trace.csv' | head -1) > $O/r04_timeline_bgr.txt
python3 tools/timeline.py $(find $O/tl_gray -name '*kernel_trace.csv' | head -1) > $O/r04_timeline_gray.txt
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_bgr/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:40]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    print(k, dict(v))
PY
