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


def oracle_of_module(frames, cfg, normal, controls, omegas, min_feat, cx, cy, model, synthetic_flow=True, hold=False):
    """optical_flow_experiments/of_module.py:78-167 written with the oracle's functions, one stream:
    re-detect (replace) when <= min_feat tracks (:83-86) -> LK (:88) -> x = [new - pix_trans, 1] in pixels (:96-102) -> u = LK flow or
    the synthetic rotational field of omega on the un-centred positions (:107-114) -> kalman.predict(control) (:122) -> legacy
    r_tilde with the predicted velocity (:125) -> keep r - (status - 1) >= T with uint8 status, i.e. tracked and r >= T (:129-131) -> A_i = [p]x / dist_i system, lstsq
    (:136-146) -> kalman.correct(-v_obs) (:152) -> old_pos = new_pos[keep] (:166).
    On <= 3 feasible points the script `continue`s without advancing the frame (:138): hold=True does the same (the device's
    ofk_fusion.hold_on_skip, one stream per context); hold=False advances the frame and lets the filter keep its prediction, which
    is what a batch of streams sharing one frame swap does (include/ofk.h, ofk_stream_step_fused).
    Returns per step (v_obs or None, filter state x, P, tracks after the step, n_old, n_kept)."""
    n = np.asarray(normal, np.float64)
    g_prev = io.gray_bgr8(frames[0])
    old = io.good_features(g_prev, cfg.max_corners, cfg.quality, cfg.min_distance, cfg.block_size).reshape(-1, 2)
    first = old.copy()
    xk, P = np.array(model.x0, np.float64), np.array(model.P0, np.float64)
    steps = []
    for t in range(1, len(frames)):
        g = io.gray_bgr8(frames[t])
        if len(old) <= min_feat:
            k = cfg.max_corners - len(old)
            old = io.good_features(g_prev, k, cfg.quality, cfg.min_distance, cfg.block_size).reshape(-1, 2) if k > 0 else np.zeros((0, 2), np.float32)
        n_old = len(old)
        if n_old:
            new, st, _ = io.lk_pyr(g_prev, g, old, cfg.win, cfg.max_level, cfg.max_count, cfg.eps, cfg.min_eig_thr)
            new = new.reshape(-1, 2); st = st.ravel().astype(np.uint8)        # cv2 hands out uint8 (of_module.py:93)
        else:
            new = np.zeros((0, 2), np.float32); st = np.zeros(0, np.uint8)
        X = new[:, 0].astype(np.float64); Y = new[:, 1].astype(np.float64)
        x3 = np.stack([X - cx, Y - cy, np.ones_like(X)], 1)
        w = np.asarray(omegas[t - 1], np.float64)
        if synthetic_flow:
            u3 = np.stack([X * Y * w[0] + (1 + X ** 2) * w[1] - Y * w[2], -(1 + Y ** 2) * w[0] + X * Y * w[1] + X * w[2], np.zeros_like(X)], 1)
        else:
            u3 = np.concatenate([new.astype(np.float64) - old.astype(np.float64), np.zeros((n_old, 1))], 1)
        xk, P = eo.kf_predict(xk, P, model.F, model.Q, model.B, np.asarray(controls[t - 1], np.float64))
        with np.errstate(divide="ignore", invalid="ignore"):
            r, dist = eo.r_tilde_legacy(x3, u3, n, xk[:3]) if n_old else (np.zeros(0), np.zeros(0))
            keep = (r - (st - np.uint8(1))) >= cfg.feas_T           # of_module.py:129 as written: uint8 status-1 wraps to 255 for a lost point
        v = None
        if keep.sum() > 3:
            v = eo.solve_of_module(x3[keep], u3[keep], dist[keep], n)[0]
            xk, P = eo.kf_correct(xk, P, model.H, model.R, -v)
        elif hold:                                               # of_module.py:138 `continue`: old_gray and old_pos stay as they are
            steps.append((None, xk.copy(), P.copy(), old.copy(), n_old, int(keep.sum())))
            continue
        old = new[keep]
        steps.append((v, xk.copy(), P.copy(), old.copy(), n_old, int(keep.sum())))
        g_prev = g
    return first, steps


def oracle_node_fused(frames, cfg, statics, imu_msgs, min_feat, radius, model=None, gps=None):
    """velocity_measurment_node with its commented-out blocks restored AND its IMU callback in the loop, one stream:
    call_imu for every message since the last frame (node:61-89: quaternion -> R, normal, dead-reckoned velocity) -> LK (:133) ->
    centre + scale (:229-235) -> solve_lgs on the tracked points with the IMU's normal / omega (:257) -> lever arm + rotation (:258)
    -> self.vel = v_uav (:261), or, with `model` (pipeline.FilterModel.ekf6), predict with the velocity increments of those
    messages and correct with +v_uav.  statics = dict(d, offset, scaling, cx, cy).  imu_msgs[t-1] = messages [M,15] before frame t.
    Returns per step (v_obs, v_uav, velocity state after the step (IMU vel or filter x), tracks, n_old, n_tracked)."""
    h, w = frames.shape[1:3]
    g_prev = io.gray_bgr8(frames[0])
    tracks = io.good_features(g_prev, cfg.max_corners, cfg.quality, cfg.min_distance, cfg.block_size).reshape(-1, 2)
    first = tracks.copy()
    state = dict(vel=np.array([0.1, 0.1, 0.1]), old_time=0.0, time_zero=0.0, first=True, rotation=np.eye(3), normal=np.array([0.0, 0, 1]),
                 ang=np.zeros(3))                                # node:182-217
    xk = P = None
    if model is not None:
        xk, P = np.array(model.x0, np.float64), np.array(model.P0, np.float64)
    steps = []
    for t in range(1, len(frames)):
        dv = np.zeros(3)
        for m in np.asarray(imu_msgs[t - 1], np.float64).reshape(-1, 15):
            v0 = state["vel"].copy()
            state = eo.imu_step(state, m[0], m[1], m[2:6], m[6:9], m[9:12], m[12:15])
            dv += state["vel"] - v0
        R = state["rotation"]; nrm = state["normal"]; om = state["ang"]
        g = io.gray_bgr8(frames[t])
        old = tracks; n_old = len(old)
        if n_old:
            new, st, _ = io.lk_pyr(g_prev, g, old, cfg.win, cfg.max_level, cfg.max_count, cfg.eps, cfg.min_eig_thr)
            new = new.reshape(-1, 2); ok = st.ravel() == 1
        else:
            new = np.zeros((0, 2), np.float32); ok = np.zeros(0, bool)
        x = (new[ok].astype(np.float64) - [statics["cx"], statics["cy"]]) * statics["scaling"]
        u = (new[ok].astype(np.float64) - old[ok]) * statics["scaling"]
        if model is not None:
            xk, P = eo.kf_predict(xk, P, model.F, model.Q, model.B, dv)
        v = vu = None
        if len(x) > 2:
            v = eo.solve_lgs_node(x, u, statics["d"], nrm, om)[0]
            vu = eo.post_solve(v, R, om, np.asarray(statics["offset"], np.float64))
            if model is not None:                              # FilterModel.ekf6(gps=True): the second velocity measurement stacked under the optical fix
                xk, P = eo.kf_correct(xk, P, model.H, model.R, vu if gps is None else np.concatenate([vu, np.asarray(gps[t - 1], np.float64)]))
            else:
                state["vel"] = vu.copy()
        tracked = new[ok]
        if n_old <= min_feat and cfg.max_corners - n_old > 0:
            mask = disc_mask(h, w, old, radius)
            newf = io.good_features(g_prev, cfg.max_corners - n_old, cfg.quality, cfg.min_distance, cfg.block_size, mask=mask).reshape(-1, 2)
            tracks = np.concatenate([tracked, newf])[:cfg.max_corners]
        else:
            tracks = tracked
        steps.append((v, vu, (xk.copy() if model is not None else state["vel"].copy()), tracks.copy(), n_old, int(ok.sum())))
        g_prev = g
    return first, steps
