#!/usr/bin/env python3
"""Soak of the double-buffered compressed ingest (GPU box): 40 batches of 64 1080p JPEG pairs through FlowPipeline.run_jpeg_batches, every
batch's records compared with the one-shot path, while a second context in the same process (a FlowStream on JPEG frames) steps every fifth batch.
   python tools/experiments/soak_ingest.py"""
import io, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from __graft_entry__ import load_package
load_package()
from of_amd import ofk, synth
from of_amd.pipeline import FlowPipeline, PipelineConfig, FlowStream
from PIL import Image
B, H, W = 64, 1080, 1920
ps = [synth.render_pair(H, W, 500 + k) for k in range(4)]
def enc(img):
    buf = io.BytesIO(); Image.fromarray(img).save(buf, "JPEG", quality=80, subsampling=2); return buf.getvalue()
sp4 = [enc(p["prev"]) for p in ps]; sn4 = [enc(p["next"]) for p in ps]
sp = [sp4[b % 4] for b in range(B)]; sn = [sn4[b % 4] for b in range(B)]
pipe = FlowPipeline(W, H, B, PipelineConfig.baseline_1080p())
sensors = ofk.make_sensors(B, scaling=ps[0]["scaling"], cx=ps[0]["cx"], cy=ps[0]["cy"])
pipe.upload_jpeg(sp, sn, sensors); ref = pipe.run()
# a second context in the same process: a camera stream on JPEG frames, interleaved with the batch ingest
fs = FlowStream(W, H, batch=1, cfg=PipelineConfig.baseline_1080p(), min_features=100, mask_radius=30)
fs.begin_jpeg([sp4[0]])
t0 = time.time(); got = []
def on_step(k):
    got.append(pipe.ctx.pairs_download(points=False)["records"].copy())
    if k % 5 == 0:
        fs.step_jpeg([sn4[k % 4]], ofk.make_sensors(1, scaling=ps[0]["scaling"], cx=ps[0]["cx"], cy=ps[0]["cy"]))
n = pipe.run_jpeg_batches(((sp, sn) for _ in range(40)), sensors, on_step=on_step)
pipe.sync()
ok = all(np.array_equal(g, ref["records"]) for g in got)
print("batches", n, "identical", ok, "seconds", round(time.time() - t0, 2))
pipe.close(); fs.close()
