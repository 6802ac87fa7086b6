R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python bench.py > gpurun_out/r04_bench_b512.json 2> gpurun_out/r04_bench_b512.err; python tools/show_bench.py gpurun_out/r04_bench_b512.json | head -1
python bench.py --config c2 > gpurun_out/r04c2_bench.json 2> gpurun_out/r04c2_bench.err; python tools/show_bench.py gpurun_out/r04c2_bench.json | head -1
python bench.py --config c4 > gpurun_out/r04c4_bench.json 2> gpurun_out/r04c4_bench.err; python tools/show_bench.py gpurun_out/r04c4_bench.json | head -1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r04_bench_b512_launcher_n1.json 2> gpurun_out/r04_launcher.err; python tools/show_bench.py gpurun_out/r04_bench_b512_launcher_n1.json | head -1
