R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; PKG=$R/drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_pipeline.py -q -x > $O/r04o_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04o_tests.log)"
cp $PKG/libofk.so /tmp/new.so
for i in 1 2 3; do for v in xold new; do
  [ $v = new ] && cp /tmp/new.so $PKG/libofk.so || cp $R/build_variants/libofk_xold.so $PKG/libofk.so
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 1 --steps 40 --warmup 3 --no-ingest --cpu-sample 0 > $O/r04o_${v}_$i.json 2>> $O/r04o.err
  python - $O/r04o_${v}_$i.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['ms_per_step'], d['config']['sharding'][:60])
PY
done; done
cp /tmp/new.so $PKG/libofk.so
