# LK with the bank-disjoint LDS layout: parity, timing (alternating with round 3's kernel = build_variants/libofk_lkold.so), LDS counters
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; PKG=$R/drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_image_parity.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -x > $O/r04d_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04d_tests.log)"
cp $PKG/libofk.so /tmp/new.so
for i in 1 2; do
  for v in lkold new; do
    [ $v = new ] && cp /tmp/new.so $PKG/libofk.so || cp $R/build_variants/libofk_lkold.so $PKG/libofk.so
    python bench.py --cpu-sample 0 --no-ingest --steps 40 > $O/r04d_${v}_$i.json 2>> $O/r04d.err
    python - $O/r04d_${v}_$i.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], 'lk alone', d['stages_isolated']['lk']['ms_per_step'], 'lk sched', d['stages']['lk']['ms_per_step'])
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for v in lkold new; do
  [ $v = new ] && cp /tmp/new.so $PKG/libofk.so || cp $R/build_variants/libofk_lkold.so $PKG/libofk.so
  rm -rf $O/pmc_lk_$v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lk_$v -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-ingest --no-overlap --streams 1 > $O/pmc_lk_$v.log 2>&1 || exit 1
  python3 - $O/pmc_lk_$v $v <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(f)):
    if r['Kernel_Name'].startswith('k_lk15q'): acc[r['Counter_Name']] += float(r['Counter_Value'])
print(sys.argv[2], {k: round(v / 5 / 1e6, 2) for k, v in acc.items()}, 'M per launch; conflict ratio', round(acc['SQ_LDS_BANK_CONFLICT'] / acc['SQ_LDS_IDX_ACTIVE'], 3))
PY
done
cp /tmp/new.so $PKG/libofk.so
