#!/usr/bin/env python3
"""What would a candidate list pruned to the strongest few thousand keys be worth?  (GPU box, timing only)
The bench workload (512 resident 1080p pairs, 500 corners) with the quality level raised step by step: while the corner count stays
at 500 the rest of the pipeline does the same work, and the step time shows what the response kernel's key stores and the
selection's passes over ~46 k candidates per image cost.
   python tools/experiments/probe_quality.py [--batch 512] [--steps 20]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    load_package()
    from of_amd import ofk, synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    B, H, W = args.batch, 1080, 1920
    truth = dict(v=(0.002, -0.0015, 0.001), omega=(0.002, -0.001, 0.003), d=1.0)
    prev, nxt, base = synth.make_batch(B, H, W, seed=2000, distinct=4, **truth)
    p0 = base[0]
    sensors = ofk.make_sensors(B, d=p0["d"], normal=p0["n"], omega=p0["omega"], scaling=p0["scaling"], cx=p0["cx"], cy=p0["cy"])
    for rep in range(2):
        for q in (0.01, 0.05, 0.1, 0.2, 0.3, 0.4):
            cfg = PipelineConfig.baseline_1080p()
            cfg.quality = q
            pipe = FlowPipeline(W, H, B, cfg)
            pipe.upload(prev, nxt, sensors)
            for _ in range(3):
                pipe.run_async()
            pipe.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pipe.run_async()
            pipe.sync()
            dt = (time.perf_counter() - t0) / args.steps
            out = pipe.ctx.pairs_download(points=False)
            rec = out["records"]
            print(f"quality {q:4.2f}: {B / dt:9.0f} pairs/s  {dt * 1e3:6.3f} ms/step  corners {rec[:, 12].mean():6.1f}  candidates {rec[:, 14].mean():8.1f}", flush=True)
            pipe.close()


if __name__ == "__main__":
    main()
