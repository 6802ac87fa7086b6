# EXPERIMENT: k_lk.hip under LLVM's max-ILP strategy with k_lk15q pinned to four waves per SIMD
cd $GRAFT_REPO_ROOT
D=$(ls -d drone*/csrc)
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I include -I $D"
run() { python3 bench.py --cpu-sample 0 --no-ingest > gpurun_out/b_ss.json 2> gpurun_out/b_ss.err; python3 -c "
import json
d=json.load(open('gpurun_out/b_ss.json'))
print('$1', d['value'], d['ms_per_step'], 'eig', d['stages_isolated']['eig']['ms_per_step'], 'lk', d['stages_isolated']['lk']['ms_per_step'])
"; }
make -s -C $D -j8 > /dev/null 2>&1; run default
cp $D/k_lk.hip /tmp/k_lk_orig.hip
sed -i "s/__global__ __launch_bounds__(64) void k_lk15q(/__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lk15q(/" $D/k_lk.hip
for s in max-ilp default; do
  X=""; [ $s != default ] && X="-mllvm -amdgpu-sched-strategy=$s"
  /opt/rocm/bin/hipcc $F $X -c $D/k_lk.hip -o $D/build/k_lk.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/../libofk.so $D/build/*.o -ldl && run "lk w4 $s"
  python3 -m pytest tests/test_gpu_image_parity.py -m gpu -q -k lk 2>&1 | tail -1
done
cp /tmp/k_lk_orig.hip $D/k_lk.hip; make -s -C $D -j8 > /dev/null 2>&1; run default
