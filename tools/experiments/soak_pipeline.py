#!/usr/bin/env python3
"""Soak of the resident pipeline (GPU box): N steps of the default schedule back to back; every `every` steps the records are downloaded
and must equal the first step's bit for bit, and the rate of every window is printed (clock / thermal drift shows as a falling rate).
  python tools/experiments/soak_pipeline.py [--steps 6000] [--every 500] [--batch 512]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6000); ap.add_argument("--every", type=int, default=500); ap.add_argument("--batch", type=int, default=512)
    a = ap.parse_args()
    load_package()
    import of_amd.ofk as ofk
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    B = a.batch
    prev, nxt, base = synth.make_batch(B, 1080, 1920, seed=2000, distinct=4, v=(0.002, -0.0015, 0.001), omega=(0.002, -0.001, 0.003), d=1.0)
    p0 = base[0]
    sensors = ofk.make_sensors(B, d=p0["d"], normal=p0["n"], omega=p0["omega"], scaling=p0["scaling"], cx=p0["cx"], cy=p0["cy"])
    pipe = FlowPipeline(1920, 1080, B, PipelineConfig.baseline_1080p())
    pipe.upload(prev, nxt, sensors)
    first = pipe.run()
    rates, bad = [], 0
    for w in range(a.steps // a.every):
        t0 = time.perf_counter()
        for _ in range(a.every):
            pipe.run_async()
        pipe.sync()
        dt = time.perf_counter() - t0
        out = pipe.ctx.pairs_download(points=True)
        same = all(np.array_equal(out[k], first[k]) for k in ("records", "counts", "prev_pts", "next_pts", "status", "err"))
        bad += 0 if same else 1
        rates.append(B * a.every / dt)
        print(f"window {w}: {rates[-1]:9.0f} pairs/s  {dt / a.every * 1e3:.4f} ms/step  records {'identical' if same else 'DIFFER'}", flush=True)
    print(f"{a.steps // a.every * a.every} steps, {bad} windows with differing records, rate first / last window {rates[0]:.0f} / {rates[-1]:.0f} pairs/s, min {min(rates):.0f}")
    pipe.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
