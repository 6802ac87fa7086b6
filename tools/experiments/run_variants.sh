#!/bin/bash
# Times bench.py with every build_variants/libofk_<name>.so copied over the product library (GPU box), two runs each, and runs the
# parity tests named in $TESTS (default: image parity + pipeline) with each.   bash tools/experiments/run_variants.sh [bench flags]
R=${GRAFT_REPO_ROOT:-/root/repo}
PKG="$R/drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd"
TESTS=${TESTS:-"tests/test_gpu_image_parity.py tests/test_gpu_pipeline.py"}
cp "$PKG/libofk.so" /tmp/libofk_product.so
for lib in "$R"/build_variants/libofk_*.so; do
    v=$(basename "$lib" .so); v=${v#libofk_}
    cp "$lib" "$PKG/libofk.so"
    (cd $R && timeout -k 10 300 python -m pytest $TESTS -q -x > gpurun_out/var_$v.test 2>&1); echo "$v tests rc=$? $(tail -1 $R/gpurun_out/var_$v.test)"
    for rep in 1 2; do
        (cd $R && timeout -k 10 200 python bench.py --cpu-sample 0 --no-ingest --steps 40 "$@" > gpurun_out/var_${v}_$rep.json 2> gpurun_out/var_${v}_$rep.err) || { echo "$v bench failed"; tail -3 $R/gpurun_out/var_${v}_$rep.err; }
        python3 - "$R/gpurun_out/var_${v}_$rep.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], "pairs/s", d["ms_per_step"], "ms/step  alone:", {k: v["ms_per_step"] for k, v in d["stages_isolated"].items() if k in ("eig", "lk", "select")})
PY
    done
done
cp /tmp/libofk_product.so "$PKG/libofk.so"
