R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
$L --master-port 29701 bench.py --gpus 1 --steps 20 --warmup 3 --streams 2 --comms 2 --batch 256 --no-ingest --cpu-sample 0 > $O/r04q_s2c2.json 2> $O/r04q_s2c2.err; echo "streams 2 comms 2 rc=$?"; python tools/show_bench.py $O/r04q_s2c2.json 2>/dev/null | head -1; grep -o '"sharding": "[^"]*"' $O/r04q_s2c2.json
$L --master-port 29702 bench.py --gpus 1 --steps 20 --warmup 3 --streams 2 --comms 1 --batch 256 --no-ingest --cpu-sample 0 > $O/r04q_s2c1.json 2> $O/r04q_s2c1.err; echo "streams 2 comms 1 rc=$?"; python tools/show_bench.py $O/r04q_s2c1.json 2>/dev/null | head -1
$L --master-port 29703 bench.py --gpus 1 --config c2 --steps 10 --warmup 2 --no-ingest --cpu-sample 0 > $O/r04q_c2.json 2> $O/r04q_c2.err; echo "c2 launcher rc=$?"; python tools/show_bench.py $O/r04q_c2.json 2>/dev/null | head -1
