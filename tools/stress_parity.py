#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): resident pair pipeline vs the CPU oracle chain on random image sizes (strip and tile
boundaries of the streaming kernels), block sizes, corner budgets and window sizes; everything compared bit for bit.
  python tools/stress_parity.py [cases] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    load_package()
    import of_amd.ofk as ofk
    from of_amd import synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    from oracle import image_oracle as io
    rng = np.random.default_rng(seed)
    widths = [64, 68, 112, 116, 120, 124, 128, 232, 236, 240, 244, 348, 352, 464, 468, 496, 500, 504, 512, 640, 700, 992, 1000, 322, 333, 479]
    t0 = time.time()
    bad = 0
    for case in range(n_cases):
        w = int(rng.choice(widths)); h = int(rng.integers(48, 300))
        bs = int(rng.choice([3, 5, 7, 7, 7, 12, 9])); win = int(rng.choice([15, 15, 15, 9, 21]))
        lvl = int(rng.integers(0, 4)); mc = int(rng.choice([10, 60, 200]))
        cfg = PipelineConfig(max_corners=mc, quality=float(rng.choice([0.01, 0.05, 0.2])), min_distance=float(rng.choice([3, 7, 10])),
                             block_size=bs, win=win, max_level=lvl, max_count=int(rng.choice([10, 20])), eps=0.03)
        B = int(rng.integers(1, 4))
        pairs = [synth.render_pair(h, w, 9000 + 17 * case + b, v=tuple(rng.normal(0, 0.004, 3)), omega=tuple(rng.normal(0, 0.003, 3)), d=1.0) for b in range(B)]
        prev = np.stack([p["prev"] for p in pairs]); nxt = np.stack([p["next"] for p in pairs])
        sensors = np.concatenate([ofk.make_sensors(1, scaling=p["scaling"], cx=p["cx"], cy=p["cy"]) for p in pairs])
        try:
            pipe = FlowPipeline(w, h, B, cfg)
            pipe.upload(prev, nxt, sensors)
            out = pipe.run()
            pipe.close()
        except ofk.OfkError as e:
            print(f"case {case}: {w}x{h} bs={bs} win={win} lvl={lvl}: library refused: {e}")
            continue
        ok = True
        for b in range(B):
            g0, g1 = io.gray_bgr8(prev[b]), io.gray_bgr8(nxt[b])
            pts = io.good_features(g0, cfg.max_corners, cfg.quality, cfg.min_distance, cfg.block_size)
            n = int(out["counts"][b])
            if pts is None or len(pts) == 0:
                ok &= n == 0
                continue
            nx, st, er = io.lk_pyr(g0, g1, pts, cfg.win, cfg.max_level, cfg.max_count, cfg.eps, cfg.min_eig_thr)
            ok &= n == len(pts) and np.array_equal(out["prev_pts"][b, :n], pts.reshape(-1, 2))
            ok &= np.array_equal(out["status"][b, :n], st.ravel())
            ok &= np.array_equal(out["next_pts"][b, :n].view(np.uint32), nx.reshape(-1, 2).view(np.uint32))
            ok &= np.array_equal(out["err"][b, :n].view(np.uint32), er.ravel().view(np.uint32))
        if not ok:
            bad += 1
        print(f"case {case:3d}: {w}x{h} B={B} bs={bs} win={win} lvl={lvl} corners<={mc} q={cfg.quality}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
