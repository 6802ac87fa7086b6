set -e
cd $GRAFT_REPO_ROOT
F=$(ls -d drone*)/csrc/k_lk.hip
cp $F /tmp/k_lk_orig.hip
for w in 0 4 5 6; do
  cp /tmp/k_lk_orig.hip $F
  if [ $w != 0 ]; then sed -i "s/__global__ __launch_bounds__(64) void k_lk15q(/__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu($w, $w))) void k_lk15q(/" $F; fi
  make -s -C $(ls -d drone*)/csrc -j8 > /dev/null 2>&1
  python3 bench.py --cpu-sample 0 --no-ingest > gpurun_out/occ_$w.json 2> gpurun_out/occ_$w.err
  python3 -c "
import json
d=json.load(open('gpurun_out/occ_$w.json'))
print('w=$w', d['value'], d['ms_per_step'], 'lk', d['stages_isolated']['lk']['ms_per_step'])
"
done
cp /tmp/k_lk_orig.hip $F
