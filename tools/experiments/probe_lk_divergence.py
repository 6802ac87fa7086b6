#!/usr/bin/env python3
"""How much does k_lk15q lose to the divergence of the four points of a wave?  (CPU only: the oracle counts the Newton steps.)

A wave tracks four consecutive corners of an image and iterates a level until the LAST of them has converged, so it pays
max-over-4 steps where the points need their own.  This script runs the oracle's LK (instrumented: orc_lk_iter_stats) on bench
frames and prints, per level, mean steps per point, mean of the per-wave maximum, and the ratio for (a) the selection's output
order (quality-sorted corners), (b) corners re-ordered by their level-L step count (the best any binning pre-pass could do at
that level), (c) by total steps (an oracle bound)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

load_package()
from of_amd import synth  # noqa: E402
from of_amd.pipeline import PipelineConfig  # noqa: E402
from oracle import image_oracle as io  # noqa: E402


def regroup_bound(tot, levels, group):
    """Per-wave maxima when the `group` points of a workgroup (group / 4 waves) are re-dealt to its waves before every level,
    sorted by the steps they needed on the previous (coarser) level — information the kernel has."""
    out = 0.0
    for st in tot:
        n = len(st) // group * group
        s = st[:n].reshape(-1, group, levels)
        total = 0.0
        prev = None
        for l in range(levels - 1, -1, -1):
            cur = s[:, :, l]
            if prev is not None:
                idx = np.argsort(prev, axis=1, kind="stable")
                cur = np.take_along_axis(cur, idx, 1); s = np.take_along_axis(s, idx[:, :, None], 1)
            total += cur.reshape(cur.shape[0], -1, 4).max(2).mean()
            prev = s[:, :, l]
        out += total
    return out / len(tot)


def main():
    cfg = PipelineConfig.baseline_1080p()
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    L = io.lib()
    tot = []; xy = []
    for s in range(pairs):
        pair = synth.render_pair(1080, 1920, 2000 + s, v=(0.002, -0.0015, 0.001), omega=(0.002, -0.001, 0.003), d=1.0)
        g0, g1 = io.gray_bgr8(pair["prev"]), io.gray_bgr8(pair["next"])
        pts = io.good_features(g0, cfg.max_corners, cfg.quality, cfg.min_distance, cfg.block_size)
        n = len(pts)
        stats = np.zeros((n, 9), np.int32)
        C.c_void_p.in_dll(L, "orc_lk_iter_stats").value = stats.ctypes.data
        io.lk_pyr(g0, g1, pts, cfg.win, cfg.max_level, cfg.max_count, cfg.eps, cfg.min_eig_thr)
        C.c_void_p.in_dll(L, "orc_lk_iter_stats").value = None
        tot.append(stats[:, :cfg.max_level + 1])
        xy.append(pts.reshape(-1, 2).copy())
    for name, order in (("selection order", None), ("sorted by coarsest-level steps", "top"), ("sorted by total steps", "sum")):
        mean_pt = np.zeros(cfg.max_level + 1); mean_wave = np.zeros(cfg.max_level + 1)
        for st in tot:
            if order == "top":
                st = st[np.argsort(st[:, -1], kind="stable")]
            elif order == "sum":
                st = st[np.argsort(st.sum(1), kind="stable")]
            n4 = len(st) // 4 * 4
            mean_pt += st[:n4].mean(0); mean_wave += st[:n4].reshape(-1, 4, st.shape[1]).max(1).mean(0)
        mean_pt /= len(tot); mean_wave /= len(tot)
        print(f"{name:34s} steps/point per level {np.round(mean_pt, 2)} = {mean_pt.sum():.2f}   per-wave max {np.round(mean_wave, 2)} = {mean_wave.sum():.2f}"
              f"   ratio {mean_wave.sum() / mean_pt.sum():.3f}")
    base = np.mean([st.mean(0).sum() for st in tot])

    def morton(x, y):
        k = np.zeros(len(x), np.int64)
        for bit in range(12):
            k |= ((x >> bit) & 1) << (2 * bit) | ((y >> bit) & 1) << (2 * bit + 1)
        return k
    for cell in (8, 16, 32, 64, 128):
        r = 0.0
        for st, p in zip(tot, xy):
            o = np.argsort(morton(p[:, 0].astype(np.int64) // cell, p[:, 1].astype(np.int64) // cell), kind="stable")
            s2 = st[o]; n4 = len(s2) // 4 * 4
            r += s2[:n4].reshape(-1, 4, s2.shape[1]).max(1).mean(0).sum()
        r /= len(tot)
        print(f"points dealt to waves in Morton order of {cell:3d}-pixel cells: per-wave max {r:.2f}   ratio {r / base:.3f}")
    for group in (16, 32, 64):
        r = regroup_bound(tot, cfg.max_level + 1, group)
        print(f"re-dealt inside groups of {group:2d} points by the previous level's steps: per-wave max {r:.2f}   ratio {r / base:.3f}")
    st = np.concatenate(tot)
    print("correlation of steps between consecutive levels (l+1 -> l):", [round(float(np.corrcoef(st[:, l + 1], st[:, l])[0, 1]), 3) for l in range(cfg.max_level)])


if __name__ == "__main__":
    main()
