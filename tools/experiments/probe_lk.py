"""Where k_lk15q's time goes: LK alone on resident pyramids of a B x 1080p batch (ofk_lk_pyr on host arrays would time the copies),
timed through the pipeline's stage brackets with max_count = 0 (set-up + error pass only), 1, 2, 3, 5, 30."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package  # noqa: E402

load_package()
import of_amd.ofk as ofk  # noqa: E402
from of_amd import synth  # noqa: E402
from of_amd.pipeline import FlowPipeline, PipelineConfig  # noqa: E402

B, H, W = 128, 1080, 1920
truth = dict(v=(0.002, -0.0015, 0.001), omega=(0.002, -0.001, 0.003), d=1.0)
prev, nxt, base = synth.make_batch(B, H, W, seed=2000, distinct=4, **truth)
p0 = base[0]
sensors = ofk.make_sensors(B, d=p0["d"], normal=p0["n"], omega=p0["omega"], scaling=p0["scaling"], cx=p0["cx"], cy=p0["cy"])
out = {}
for mc in (0, 1, 2, 3, 5, 30):
    cfg = PipelineConfig.baseline_1080p()
    cfg.max_count = mc
    pipe = FlowPipeline(W, H, B, cfg, streams=1)
    pipe.ctx.set_overlap(False)
    pipe.upload(prev, nxt, sensors)
    pipe.run_async(); pipe.sync()
    pipe.ctx.profile_read(); pipe.ctx.profile_enable(0x7f)
    for _ in range(5):
        pipe.run_async()
    pipe.sync()
    prof = pipe.ctx.profile_read()
    out[mc] = round(prof["lk"][0] / prof["lk"][1], 4)
    pipe.ctx.profile_enable(0)
    del pipe
print(json.dumps({"lk_ms_per_launch_of_%d_pairs_by_max_count" % B: out}))
