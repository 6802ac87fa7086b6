#!/bin/bash
# Per-kernel times of the JPEG decoder for several chunk sizes (GPU box):  bash tools/experiments/jpeg_chunks.sh [batch] [chunks...]
B=${1:-256}; shift; CH=${@:-256 512 1024}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in $CH; do
  rm -rf $O/jc_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/jc_$c -- python3 $R/tools/bench_jpeg.py --batch $B --reps 3 --chunk $c > $O/jc_$c.log 2>&1 || { tail -5 $O/jc_$c.log; exit 1; }
  echo "== chunk $c"; grep matches_oracle $O/jc_$c.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('jpeg_upload_ms_per_call','decode_ms_per_call','jpeg_double_buffered_ms_per_call','matches_oracle')})"
  python3 - <<PY
import csv, glob
f = glob.glob("$O/jc_$c/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    if "jpeg" in r["Name"]:
        print("   %-22s calls %5s  mean %9.1f us  total %9.2f ms" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6)); tot += float(r["TotalDurationNs"]) / 1e6
print("   decoder kernels total %.2f ms in the run" % tot)
PY
done
