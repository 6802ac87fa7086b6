#!/usr/bin/env python3
"""Throughput of the resident pair pipeline on the other BASELINE.json configurations (not the bench contract's line):
C3 = 1024 x 640x480 pairs (+ a 6-state Kalman update per pair), C4 = 1080p camera streams frame by frame (FlowStream),
C5 = 3840x2160 pairs, 2000 corners, 5-level pyramid.
  python tools/bench_configs.py [c3] [c4] [c5]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


RESULTS = []


def run(name, w, h, batch, cfg, steps, kf=False):
    import of_amd.ofk as ofk
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline
    prev, nxt, base = synth.make_batch(batch, h, w, seed=77, distinct=4)
    sensors = ofk.make_sensors(batch, scaling=base[0]["scaling"], cx=base[0]["cx"], cy=base[0]["cy"])
    pipe = FlowPipeline(w, h, batch, cfg, streams=int(os.environ.get("OFK_STREAMS", "1")))     # one stage chain + auxiliary stream, like bench.py
    pipe.upload(prev, nxt, sensors)
    if os.environ.get("OFK_NO_OVERLAP"):
        pipe.ctx.set_overlap(False)                              # every kernel alone on the chip (profiling)
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
    res = {"config": name, "frame": f"{w}x{h}", "pairs_per_step": batch, "corners_mean": float(np.mean(out["counts"])), "pairs_per_s": round(batch * steps / dt, 1),
           "ms_per_step": round(dt / steps * 1e3, 4), "stage_ms_per_step_under_overlap": {s: round(prof[s][0] / steps, 4) for s in ofk.STAGES if prof[s][1]}}
    if kf:                                                      # the 6-state filter of every pair, state resident, queued behind each step
        from of_amd.pipeline import FilterModel
        pipe.ctx.filter_configure(FilterModel.ekf6(), batch)
        for _ in range(2):
            pipe.run_async(); pipe.ctx.pairs_filter_step(batch)
        pipe.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.run_async(); pipe.ctx.pairs_filter_step(batch)
        pipe.sync()
        dtk = time.perf_counter() - t0
        res["with_resident_ekf6_per_pair"] = {"pairs_per_s": round(batch * steps / dtk, 1), "ms_per_step": round(dtk / steps * 1e3, 4)}
    RESULTS.append(res)
    print(json.dumps(res), flush=True)
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
    res = {"config": name, "frame": f"{w}x{h}", "streams": streams, "ms_per_step_host_bgr_frames": round(dt * 1e3, 4), "frames_per_s": round(streams / dt, 1),
           "tracks_per_stream": float(np.mean(counts)), "camera_period_ms_at_30fps": round(1e3 / 30, 2)}
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
        res["ms_per_step_jpeg_frames"] = round(dt * 1e3, 4); res["jpeg_kB_per_frame"] = round(float(np.mean([len(s) for s in j0])) / 1e3, 1)
    except ImportError:
        pass
    fs.close()
    # the same loop with the filters resident (IMU dead-reckoning + 6-state filter, ofk_stream_step_fused)
    from of_amd.pipeline import FusionConfig
    fs = FlowStream(w, h, batch=streams, cfg=cfg, min_features=cfg.max_corners // 2, mask_radius=10, fusion=FusionConfig.ekf6())
    fs.begin(f0)
    msgs = np.zeros((streams, 2, 15)); msgs[:, :, 5] = 1.0; msgs[:, :, 14] = 9.81
    msgs[:, 0, 0] = 100; msgs[:, 1, 0] = 100; msgs[:, 1, 1] = 2e7
    fs.push_imu(msgs); fs.step_fused(f1, sensors)
    t0 = time.perf_counter()
    for k in range(frames):
        msgs[:, :, 0] += 1
        fs.push_imu(msgs)
        rec, fused, tracks, counts = fs.step_fused(f0 if k & 1 else f1, sensors)
    dt = (time.perf_counter() - t0) / frames
    res["ms_per_step_fused_imu_ekf6"] = round(dt * 1e3, 4)
    fs.close()
    RESULTS.append(res)
    print(json.dumps(res), flush=True)


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
    out = os.environ.get("OFK_CONFIGS_JSON")
    if out:
        json.dump({"_note": "tools/bench_configs.py on one MI355X: the other BASELINE.json configurations (bench.py measures configs[1]); "
                            "two free-running slices like bench.py; C3 = configs[2], C4 = configs[3] (1 and 8 streams in one context), C5 = configs[4]'s pair workload",
                   "results": RESULTS}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
