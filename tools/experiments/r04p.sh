R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_estimation_parity.py -q -x > $O/r04p_tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/r04p_tests.log)"
python - <<'PY'
# configs[4]'s Monte-Carlo sweep at its own size: 2000 points, 4096 trials per step - host noise (numpy + upload) against device noise
import time, numpy as np, sys
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package()
import of_amd.simulation as sim
rng = np.random.default_rng(1)
pts = rng.uniform(-1.2, 1.2, (2000, 2))
args = (pts, [1.0, 1, 1], [1.0, 1, 1], 1.0, [0.0, 0, 1], [0.02, 0, 0.205])
for name, kw in (("host noise (numpy generator, 262 MB per step uploaded)", dict(generator=np.random.default_rng(3))), ("device noise (Philox4x32-10 in the kernel)", dict(device_seed=7))):
    sim.sweep("flow_errors", *args, k=100, trials=4096, steps=[5], **kw)
    t0 = time.perf_counter()
    out = sim.sweep("flow_errors", *args, k=100, trials=4096, steps=list(range(10, 20)), **kw)
    dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt * 1e3:.2f} ms per sweep step of 4096 trials x 2000 points = {4096 / dt:.0f} trials/s; mean v of step 15: {out[15:18] if False else out[:30].reshape(10,3)[5]}")
PY
