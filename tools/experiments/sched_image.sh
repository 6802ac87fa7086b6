# EXPERIMENT: scheduling strategies for k_image.hip (gray, pyramid pass)
cd $GRAFT_REPO_ROOT
D=$(ls -d drone*/csrc)
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I include -I $D"
run() { python3 bench.py --cpu-sample 0 --no-ingest > gpurun_out/b_ss.json 2> gpurun_out/b_ss.err; python3 -c "
import json
d=json.load(open('gpurun_out/b_ss.json'))
print('$1', d['value'], d['ms_per_step'], 'gray', d['stages_isolated']['gray']['ms_per_step'], 'pyr', d['stages_isolated']['pyr']['ms_per_step'])
"; }
make -s -C $D -j8 > /dev/null 2>&1; run default
for s in max-ilp max-memory-clause; do
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=$s -c $D/k_image.hip -o $D/build/k_image.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/../libofk.so $D/build/*.o -ldl && run "image:$s" && run "image:$s"
done
touch $D/k_image.hip; make -s -C $D -j8 > /dev/null 2>&1; run default
