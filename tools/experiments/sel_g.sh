cd $GRAFT_REPO_ROOT
D=$(ls -d drone*/csrc)
for gcount in 16 32 68; do
sed -i "s/^#define SEL_G .*/#define SEL_G $gcount/" $D/k_corners.hip
make -s -C $D -j8 > /dev/null 2>&1
python3 -m pytest tests/test_gpu_image_parity.py -m gpu -q -k "good_features or select" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/ps && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ps -o p -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --no-ingest --no-overlap --streams 1 --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/ps.log 2>&1; cd $GRAFT_REPO_ROOT
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/ps/p_kernel_stats.csv')):
    if 'select' in r['Name']: print('SEL_G $gcount', r['Name'][:16], round(float(r['AverageNs'])/1e3,1))
"
done
sed -i "s/^#define SEL_G .*/#define SEL_G 16/" $D/k_corners.hip
