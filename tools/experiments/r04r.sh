R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_pipeline.py -q -x > $O/r04r_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04r_tests.log)"
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
for i in 1 2; do
$L --master-port $((29710 + i)) bench.py --gpus 1 --config c2 --steps 20 --warmup 3 --no-ingest --cpu-sample 0 > $O/r04r_c2_$i.json 2> $O/r04r_c2.err; python tools/show_bench.py $O/r04r_c2_$i.json 2>/dev/null | head -1
python bench.py --config c2 --steps 20 --warmup 3 --no-ingest --cpu-sample 0 > $O/r04r_c2plain_$i.json 2>> $O/r04r_c2.err; python tools/show_bench.py $O/r04r_c2plain_$i.json 2>/dev/null | head -1
done
$L --master-port 29720 bench.py --gpus 1 --steps 20 --warmup 3 --streams 2 --comms 1 --batch 256 --no-ingest --cpu-sample 0 > $O/r04r_s2c1.json 2> $O/r04r_s2c1.err; python tools/show_bench.py $O/r04r_s2c1.json 2>/dev/null | head -1
