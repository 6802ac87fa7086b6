#!/bin/bash
# Response-kernel key stores: ballot + mbcnt ranks (base) against slots from an LDS counter (atom, -DOFK_KEYS_ATOM=1).
#   tools/variants.sh build k_corners.hip "base:-DOFK_KEYS_ATOM=0" "atom:-DOFK_KEYS_ATOM=1"      (build container)
#   bash tools/experiments/eig_keys.sh                                                            (GPU box)
# Each variant: the corner parity tests, then bench.py (serial kernel times in stages_isolated).
R=${GRAFT_REPO_ROOT:-/root/repo}
PKG="$R/drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd"
cp "$PKG/libofk.so" /tmp/libofk_product.so
for v in base atom; do
    cp "$R/build_variants/libofk_$v.so" "$PKG/libofk.so"
    (cd $R && timeout -k 10 300 python -m pytest tests/test_gpu_image_parity.py tests/test_gpu_pipeline.py -q -x > gpurun_out/eigkeys_$v.test 2>&1); echo "$v tests rc=$? $(tail -1 $R/gpurun_out/eigkeys_$v.test)"
    for rep in 1 2; do
        (cd $R && timeout -k 10 200 python bench.py --cpu-sample 0 --no-ingest --steps 40 > gpurun_out/eigkeys_${v}_$rep.json 2> gpurun_out/eigkeys_${v}_$rep.err) || { echo "$v bench failed"; tail -3 $R/gpurun_out/eigkeys_${v}_$rep.err; }
        python3 - "$R/gpurun_out/eigkeys_${v}_$rep.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], "pairs/s", d["ms_per_step"], "ms/step  eig alone", d["stages_isolated"]["eig"]["ms_per_step"], " select alone", d["stages_isolated"]["select"]["ms_per_step"])
PY
    done
done
cp /tmp/libofk_product.so "$PKG/libofk.so"
