# EXPERIMENT: further scheduler switches on top of max-ilp for k_corners.hip
cd $GRAFT_REPO_ROOT
D=$(ls -d drone*/csrc)
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -I include -I $D -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp"
run() { python3 bench.py --cpu-sample 0 --no-ingest > gpurun_out/b_ss.json 2> gpurun_out/b_ss.err; python3 -c "
import json
d=json.load(open('gpurun_out/b_ss.json'))
print('$1', d['value'], d['ms_per_step'], 'eig', d['stages_isolated']['eig']['ms_per_step'], 'select', d['stages_isolated']['select']['ms_per_step'])
"; }
make -s -C $D -j8 > /dev/null 2>&1; run base
for x in "-mllvm -enable-post-misched=false" "-mllvm -misched-cluster=false" "-O2" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -amdgpu-schedule-metric-bias=50"; do
  /opt/rocm/bin/hipcc $F $x -c $D/k_corners.hip -o $D/build/k_corners.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/../libofk.so $D/build/*.o -ldl && run "corners: $x"
done
touch $D/k_corners.hip; make -s -C $D -j8 > /dev/null 2>&1; run base
