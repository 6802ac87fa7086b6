#!/usr/bin/env python3
"""Throughput of the resident pair pipeline on the other BASELINE.json configurations (not the bench contract's line):
C3 = 1024 x 640x480 pairs (+ a 6-state Kalman update per pair), C4 = 1080p camera streams frame by frame (FlowStream),
C5 = 3840x2160 pairs, 2000 corners, 5-level pyramid.
  python tools/bench_configs.py [c3] [c4] [c5]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def run(name, w, h, batch, cfg, steps, kf=False):
    import of_amd.ofk as ofk
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline
    prev, nxt, base = synth.make_batch(batch, h, w, seed=77, distinct=4)
    sensors = ofk.make_sensors(batch, scaling=base[0]["scaling"], cx=base[0]["cx"], cy=base[0]["cy"])
    pipe = FlowPipeline(w, h, batch, cfg, streams=int(os.environ.get("OFK_STREAMS", "2")))     # two free-running slices, like bench.py
    pipe.upload(prev, nxt, sensors)
    for _ in range(2):
        pipe.run_async()
    pipe.sync()
    pipe.ctx.profile_read(); pipe.ctx.profile_enable(0x7f)
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.run_async()
    pipe.sync()
    dt = time.perf_counter() - t0
    prof = pipe.ctx.profile_read()
    out = pipe.ctx.pairs_download(points=False)
    line = f"{name}: {w}x{h} B={batch} corners={np.mean(out['counts']):.0f}: {batch * steps / dt:.0f} pairs/s ({dt / steps * 1e3:.3f} ms/step)  " + \
        " ".join(f"{s}={prof[s][0] / steps:.3f}" for s in ofk.STAGES if prof[s][1])
    if kf:                                                      # of_module.py:63-76 shaped filter with 6 states on every pair's velocity
        ns, nm = 6, 3
        F = np.eye(ns); F[:3, 3:] = np.eye(3) / 30.0
        H = np.zeros((nm, ns)); H[:, :3] = np.eye(3)
        x = np.zeros((batch, ns)); P = np.tile(np.eye(ns), (batch, 1, 1))
        z = out["records"][:, 0:3].copy()
        t0 = time.perf_counter()
        for _ in range(steps):
            x, P = pipe.ctx.kf_predict_update(F, H, 1e-4 * np.eye(ns), 1e-2 * np.eye(nm), x, P, z=z)
        line += f"  | kalman(6 states, host buffers): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per {batch} filters"
    print(line, flush=True)
    pipe.close()


def run_stream(name, w, h, streams, cfg, frames):
    """C4: `streams` camera streams, one new frame per stream and step (host BGR in, PCIe included): LK from the resident tracks,
    velocity, status filter, masked re-detection, frame swap - pipeline.FlowStream / ofk_stream_step."""
    import of_amd.ofk as ofk
    from of_amd import synth
    from of_amd.pipeline import FlowStream
    pairs = [synth.render_pair(h, w, 500 + k) for k in range(streams)]
    f0 = np.stack([p["prev"] for p in pairs]); f1 = np.stack([p["next"] for p in pairs])
    sensors = np.concatenate([ofk.make_sensors(1, scaling=p["scaling"], cx=p["cx"], cy=p["cy"]) for p in pairs])
    fs = FlowStream(w, h, batch=streams, cfg=cfg, min_features=cfg.max_corners // 2, mask_radius=10)
    fs.begin(f0)
    fs.step(f1, sensors)
    t0 = time.perf_counter()
    for k in range(frames):
        rec, tracks, counts = fs.step(f0 if k & 1 else f1, sensors)
    dt = (time.perf_counter() - t0) / frames
    print(f"{name}: {streams} stream(s) {w}x{h}: {dt * 1e3:.3f} ms per step = {streams / dt:.0f} frames/s, tracks per stream {np.mean(counts):.0f} "
          f"(a 30 fps camera leaves {1e3 / 30:.1f} ms per frame)", flush=True)
    try:                                                        # the same loop on compressed frames (what the node receives), if an encoder is around
        import io
        from PIL import Image

        def enc(img):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, "JPEG", quality=80)
            return buf.getvalue()
        j0 = [enc(f) for f in f0]; j1 = [enc(f) for f in f1]
        fs.begin_jpeg(j0)
        fs.step_jpeg(j1, sensors)
        t0 = time.perf_counter()
        for k in range(frames):
            rec, tracks, counts = fs.step_jpeg(j0 if k & 1 else j1, sensors)
        dt = (time.perf_counter() - t0) / frames
        print(f"{name} (JPEG frames, {np.mean([len(s) for s in j0]) / 1e3:.0f} kB each): {dt * 1e3:.3f} ms per step = {streams / dt:.0f} frames/s", flush=True)
    except ImportError:
        pass
    fs.close()


def main():
    load_package()
    from of_amd.pipeline import PipelineConfig
    which = sys.argv[1:] or ["c3", "c4", "c5"]
    if "c3" in which:
        run("C3", 640, 480, 1024, PipelineConfig(max_corners=500, quality=0.01, min_distance=10, block_size=7, win=15, max_level=3), 10, kf=True)
    if "c4" in which:
        cfg4 = PipelineConfig(max_corners=500, quality=0.01, min_distance=10, block_size=7, win=15, max_level=3)
        run_stream("C4", 1920, 1080, 1, cfg4, 60)
        run_stream("C4x8", 1920, 1080, 8, cfg4, 30)
    if "c5" in which:
        run("C5", 3840, 2160, 32, PipelineConfig(max_corners=2000, quality=0.01, min_distance=10, block_size=7, win=15, max_level=5), 10)


if __name__ == "__main__":
    main()
