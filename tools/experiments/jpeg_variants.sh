#!/bin/bash
# Per-kernel times of the JPEG decoder for every library in build_variants/ (GPU box):  bash tools/experiments/jpeg_variants.sh [batch]
B=${1:-256}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; PKG=$R/drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd
cd /tmp && export TMPDIR=/tmp
for lib in $R/build_variants/libofk_*.so; do
  name=$(basename $lib .so); name=${name#libofk_}
  cp $lib $PKG/libofk.so
  rm -rf $O/jv_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/jv_$name -- python3 $R/tools/bench_jpeg.py --batch $B --reps 3 > $O/jv_$name.log 2>&1 || { tail -5 $O/jv_$name.log; exit 1; }
  echo "== $name"; grep matches_oracle $O/jv_$name.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('jpeg_upload_ms_per_call','decode_ms_per_call','jpeg_double_buffered_ms_per_call','matches_oracle')})"
  python3 - <<PY
import csv, glob
f = glob.glob("$O/jv_$name/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "jpeg" in r["Name"]: print("   %-22s calls %5s  mean %9.1f us  total %9.2f ms" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
