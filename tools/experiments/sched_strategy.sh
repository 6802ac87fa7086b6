# EXPERIMENT: LLVM AMDGPU scheduling strategies for the two VALU-bound translation units (rebuilds on the GPU box, restores at the end)
cd $GRAFT_REPO_ROOT
D=$(ls -d drone*/csrc)
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I include -I $D"
run() { python3 bench.py --cpu-sample 0 --no-ingest > gpurun_out/b_ss.json 2> gpurun_out/b_ss.err; python3 -c "
import json
d=json.load(open('gpurun_out/b_ss.json'))
print('$1', d['value'], d['ms_per_step'], 'eig', d['stages_isolated']['eig']['ms_per_step'], 'lk', d['stages_isolated']['lk']['ms_per_step'])
"; }
make -s -C $D -j8 > /dev/null 2>&1; run default
for s in max-ilp max-memory-clause iterative-maxocc; do
  /opt/rocm/bin/hipcc $F -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=$s -c $D/k_corners.hip -o $D/build/k_corners.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/../libofk.so $D/build/*.o -ldl && run "corners:$s"
done
make -s -C $D -j8 > /dev/null 2>&1; touch $D/k_corners.hip; make -s -C $D -j8 > /dev/null 2>&1
for s in max-ilp max-memory-clause iterative-maxocc; do
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=$s -c $D/k_lk.hip -o $D/build/k_lk.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/../libofk.so $D/build/*.o -ldl && run "lk:$s"
done
touch $D/k_lk.hip; make -s -C $D -j8 > /dev/null 2>&1; run default
