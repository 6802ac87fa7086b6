"""The loop of velocity_measurment_node:92-177 (commented-out blocks restored) written with the CPU oracle's functions:
the checker the video-stream GPU tests compare `FlowStream` against, frame by frame.  Test infrastructure only."""
import numpy as np

from oracle import image_oracle as io, estimation_oracle as eo


def disc_mask(h, w, pts, radius):
    m = np.ones((h, w), np.uint8)
    for x, y in pts:
        cx, cy = int(x), int(y)
        y0, y1 = max(0, cy - radius), min(h, cy + radius + 1); x0, x1 = max(0, cx - radius), min(w, cx + radius + 1)
        if y0 < y1 and x0 < x1:
            yy, xx = np.ogrid[y0:y1, x0:x1]
            m[y0:y1, x0:x1][(yy - cy) ** 2 + (xx - cx) ** 2 <= radius * radius] = 0
    return m


def oracle_stream(frames, cfg, sensors, min_feat, radius):
    """node:117-175 (restored): returns per step (v_obs or None, tracks after the step, n_old, n_tracked)."""
    h, w = frames.shape[1:3]
    g_prev = io.gray_bgr8(frames[0])
    tracks = io.good_features(g_prev, cfg.max_corners, cfg.quality, cfg.min_distance, cfg.block_size).reshape(-1, 2)
    first = tracks.copy()
    steps = []
    for t in range(1, len(frames)):
        g = io.gray_bgr8(frames[t])
        old = tracks; n_old = len(old)
        if n_old:
            new, st, _ = io.lk_pyr(g_prev, g, old, cfg.win, cfg.max_level, cfg.max_count, cfg.eps, cfg.min_eig_thr)
            new = new.reshape(-1, 2); ok = st.ravel() == 1
        else:
            new = np.zeros((0, 2), np.float32); ok = np.zeros(0, bool)
        sr = sensors
        x = (new[ok].astype(np.float64) - [sr[20], sr[21]]) * sr[19]; u = (new[ok].astype(np.float64) - old[ok]) * sr[19]
        v = eo.solve_lgs_node(x, u, sr[0], sr[1:4], sr[4:7])[0] if len(x) >= 3 else None
        tracked = new[ok]
        if n_old <= min_feat and cfg.max_corners - n_old > 0:
            mask = disc_mask(h, w, old, radius)
            newf = io.good_features(g_prev, cfg.max_corners - n_old, cfg.quality, cfg.min_distance, cfg.block_size, mask=mask).reshape(-1, 2)
            tracks = np.concatenate([tracked, newf])[:cfg.max_corners]
        else:
            tracks = tracked
        steps.append((v, tracks.copy(), n_old, int(ok.sum())))
        g_prev = g
    return first, steps
