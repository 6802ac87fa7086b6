#!/usr/bin/env python3
"""Which chain bounds the step?  (GPU box, timing only)  The bench workload with LK cut short (max_count 20 / 3 / 1 / 0 Newton steps) and with
fewer corners (500 / 250 / 100): the main chain (response -> selection -> LK -> solve) gets shorter, the auxiliary one (gray, pyramids) stays.
   python tools/experiments/probe_chain.py [--batch 512] [--steps 20]"""
import argparse, os, sys, time
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
        for mc, corners in ((20, 500), (3, 500), (1, 500), (20, 250), (20, 100), (1, 100)):
            cfg = PipelineConfig.baseline_1080p()
            cfg.max_count = mc; cfg.max_corners = corners
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
            print(f"max_count {mc:2d} corners {corners:3d}: {B / dt:9.0f} pairs/s  {dt * 1e3:6.3f} ms/step", flush=True)
            pipe.close()


if __name__ == "__main__":
    main()
