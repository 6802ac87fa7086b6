"""GPU: the filters inside the resident stream loop (ofk_stream_step_fused; SURVEY.md §8(e) "the per-stream filter stays on the GPU
that owns the stream").

  * of_module.py:78-167 end to end for 10 frames, two streams: KF predict -> legacy r_tilde with the predicted velocity ->
    [p]x / dist_i system -> correct(-v_obs), tracks := kept points, replace-re-detection — against the same loop written with
    the oracle's functions (tests/stream_oracle.py::oracle_of_module); with the synthetic rotational flow of :113-114 and with
    the LK flow the script's TODO asks for.
  * the node's loop with its IMU callback resident (call_imu between frames, dead-reckoned velocity as r_tilde prior, self.vel =
    v_uav) and the same loop with the build-defined 6-state filter (FilterModel.ekf6) in place of the overwrite.
  * the 3-state filter through the fused path is bit-identical to the stand-alone ofk_kf_predict_update (the closed-form KAT's kernel).
"""
import numpy as np
import pytest

from stream_oracle import oracle_of_module, oracle_node_fused

pytestmark = pytest.mark.gpu


def make_imu_msgs(rng, t0, n, tilt=0.02):
    """n messages 20 ms apart from time t0 (seconds): a slightly tilted, slowly rotating vehicle."""
    out = np.zeros((n, 15))
    for k in range(n):
        t = t0 + 0.02 * (k + 1)
        ax = rng.normal(0, tilt, 3)
        q = np.array([ax[0] / 2, ax[1] / 2, ax[2] / 2, 1.0]); q /= np.linalg.norm(q)
        out[k] = [int(t), int((t - int(t)) * 1e9), *q, *rng.normal(0, 0.002, 3), 1e-4, 2e-4, 3e-4, *(rng.normal(0, 0.05, 3) + [0, 0, 9.81])]
    return out


@pytest.mark.parametrize("synthetic", [True, False])
def test_of_module_loop_ten_frames_matches_oracle_composition(pkg, ofk, synthetic):
    from of_amd import synth
    from of_amd.of_library import pix_trans
    from of_amd.pipeline import FlowStream, PipelineConfig, FusionConfig
    h, w, nf, B = 480, 640, 11, 2
    cfg = PipelineConfig.of_module()
    cfg.max_corners = 60; cfg.quality = 0.05; cfg.block_size = 7; cfg.min_distance = 12       # enough corners on the synthetic texture
    if not synthetic:
        cfg.feas_T = -0.5                                        # real flow: r scatters around its mean; keep most points (T is a parameter, of_module.py:55)
    seqs = [synth.render_sequence(h, w, 900 + b, nf, v=(0.004 + 0.002 * b, -0.003, 0.002), omega=(0.001, -0.002, 0.003 * (b + 1)), d=1.0) for b in range(B)]
    frames = np.stack([s[0] for s in seqs])
    rng = np.random.default_rng(77)
    controls = rng.normal(0, 0.01, (nf - 1, B, 3)); omegas = rng.normal(0, 0.01, (nf - 1, B, 3))        # of_module.py:111,122 draw them per frame
    cx, cy = pix_trans((480, 640))                               # (240, 320), applied to (x, y) in that order as the script does (:100-102)
    normal = np.array([0.0, 0.0, 1.0])
    fusion = FusionConfig.of_module(synthetic_flow=synthetic)
    min_feat = 40 if synthetic else 10                           # of_module.py:83 uses 10; 40 makes the replace-re-detection fire on these clips
    fs = FlowStream(w, h, batch=B, cfg=cfg, min_features=min_feat, mask_radius=10, fusion=fusion)
    tracks, counts = fs.begin(frames[:, 0])
    refs = [oracle_of_module(frames[b], cfg, normal, controls[:, b], omegas[:, b], min_feat, cx, cy, fusion.model, synthetic) for b in range(B)]
    for b in range(B):
        assert counts[b] == len(refs[b][0]) and np.array_equal(tracks[b, :counts[b]], refs[b][0])
    solved_any = False
    for t in range(1, nf):
        sensors = np.concatenate([ofk.make_sensors(1, d=1.0, normal=normal, omega=omegas[t - 1, b], scaling=1.0, cx=cx, cy=cy) for b in range(B)])
        sensors[:, 25:28] = controls[t - 1]
        rec, fused, tracks, counts = fs.step_fused(frames[:, t], sensors)
        for b in range(B):
            v, xk, P, tr, n_old, n_keep = refs[b][1][t - 1]
            assert rec[b, 12] == n_old and rec[b, 11] == n_keep and counts[b] == len(tr), (t, b, rec[b, 11:14], n_old, n_keep, len(tr))
            assert np.array_equal(tracks[b, :counts[b]].view(np.uint32), tr.astype(np.float32).view(np.uint32)), (t, b)
            if v is not None:
                solved_any = True
                np.testing.assert_allclose(rec[b, :3], v, rtol=1e-7, atol=1e-12)
                assert rec[b, 15] == 1 and fused[b, 7] == 1
            else:
                assert rec[b, 15] == 0 and rec[b, 4] == 0
            np.testing.assert_allclose(fused[b, :3], xk, rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(fused[b, 6], np.trace(P), rtol=1e-10)
    gx, gP = fs.ctx.filter_state(B)
    for b in range(B):
        np.testing.assert_allclose(gx[b], refs[b][1][-1][1], rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(gP[b], refs[b][1][-1][2], rtol=1e-10, atol=1e-14)
    assert solved_any
    fs.close()


@pytest.mark.parametrize("use_ekf", [False, True])
def test_node_loop_with_resident_imu_and_filter(pkg, ofk, use_ekf):
    from of_amd import synth
    from of_amd.pipeline import FlowStream, PipelineConfig, FusionConfig
    h, w, nf, B = 480, 640, 8, 2
    cfg = PipelineConfig(max_corners=120, quality=0.02, min_distance=10, block_size=7)
    seqs = [synth.render_sequence(h, w, 940 + b, nf, v=(0.004, -0.003 + 0.002 * b, 0.002), omega=(0.002, 0.001, -0.003), d=1.0) for b in range(B)]
    frames = np.stack([s[0] for s in seqs]); info = seqs[0][1]
    statics = dict(d=1.0, offset=(0.0, 0.0, 0.1), scaling=info["scaling"], cx=info["cx"], cy=info["cy"])
    rng = np.random.default_rng(5)
    msgs = np.stack([[make_imu_msgs(rng, 100.0 + 0.1 * t + 7 * b, 3) for b in range(B)] for t in range(nf - 1)])     # [nf-1, B, 3, 15]
    fusion = FusionConfig.ekf6(dt=0.1) if use_ekf else FusionConfig.node()
    fs = FlowStream(w, h, batch=B, cfg=cfg, min_features=115, mask_radius=15, fusion=fusion)
    tracks, counts = fs.begin(frames[:, 0])
    refs = [oracle_node_fused(frames[b], cfg, statics, msgs[:, b], 115, 15, fusion.model if use_ekf else None) for b in range(B)]
    sensors = ofk.make_sensors(B, d=statics["d"], offset=statics["offset"], scaling=statics["scaling"], cx=statics["cx"], cy=statics["cy"])
    for t in range(1, nf):
        fs.push_imu(msgs[t - 1])                                 # three messages per stream since the last frame
        rec, fused, tracks, counts = fs.step_fused(frames[:, t], sensors)
        for b in range(B):
            v, vu, vel, tr, n_old, n_tr = refs[b][1][t - 1]
            assert rec[b, 12] == n_old and rec[b, 13] == n_tr and counts[b] == len(tr)
            assert np.array_equal(tracks[b, :counts[b]].view(np.uint32), tr.astype(np.float32).view(np.uint32)), (t, b)
            np.testing.assert_allclose(rec[b, :3], v, rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(rec[b, 8:11], vu, rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(fused[b, :len(vel)] if use_ekf else fused[b, :3], vel, rtol=1e-8, atol=1e-12)
    st, dv = fs.ctx.imu_state(B)
    if not use_ekf:                                              # node:261: the dead-reckoning state carries the last optical fix
        for b in range(B):
            np.testing.assert_allclose(st[b, 0:3], refs[b][1][-1][2], rtol=1e-8, atol=1e-12)
    assert np.all(dv == 0)                                       # a step starts a new dead-reckoning interval
    # ragged message counts: a stream with fewer messages only applies those
    before, _ = fs.ctx.imu_state(B)
    extra = np.stack([make_imu_msgs(rng, 300.0 + b, 4) for b in range(B)])
    fs.push_imu(extra, np.array([4, 1], np.int32))
    after, _ = fs.ctx.imu_state(B)
    assert after[0, 3] != before[0, 3] and abs(after[1, 3] - (300.0 + 1 + 0.02 - after[1, 4])) < 1e-6
    fs.close()


def test_fused_three_state_filter_is_bit_identical_to_the_stand_alone_kernel(pkg, ofk, gpu_ctx):
    """The reference filter (F = B = H = I, of_module.py:63-76) through ofk_stream_step_fused == ofk_kf_predict_update fed with the
    same controls and measurements, bit for bit, over several frames (3-state must stay what the closed-form KAT pins)."""
    from of_amd import synth
    from of_amd.of_library import pix_trans
    from of_amd.pipeline import FlowStream, PipelineConfig, FusionConfig, FilterModel
    h, w, nf = 240, 320, 6
    cfg = PipelineConfig.of_module(); cfg.max_corners = 60; cfg.quality = 0.05; cfg.block_size = 7; cfg.min_distance = 8; cfg.max_level = 2
    frames, info = synth.render_sequence(h, w, 77, nf, v=(0.004, -0.003, 0.002), omega=(0.002, 0.001, -0.003), d=1.0)
    fusion = FusionConfig.of_module(synthetic_flow=True)
    fs = FlowStream(w, h, batch=1, cfg=cfg, min_features=5, mask_radius=10, fusion=fusion)
    fs.begin(frames[0][None])
    m = FilterModel.kf3()
    x, P = m.x0.copy(), m.P0.copy()
    rng = np.random.default_rng(3)
    cx, cy = pix_trans((240, 320))
    for t in range(1, nf):
        ctrl = rng.normal(0, 0.01, 3); om = rng.normal(0, 0.01, 3)
        sensors = ofk.make_sensors(1, d=1.0, omega=om, scaling=1.0, cx=cx, cy=cy); sensors[:, 25:28] = ctrl
        rec, fused, tracks, counts = fs.step_fused(frames[t][None], sensors)
        x, P = gpu_ctx.kf_predict_update(m.F, m.H, m.Q, m.R, x, P, B=m.B, u=ctrl, z=(-rec[0, :3] if rec[0, 15] else None))
        gx, gP = fs.ctx.filter_state(1)
        assert np.array_equal(gx[0].view(np.uint64), x.view(np.uint64)) and np.array_equal(gP[0].view(np.uint64), P.view(np.uint64)), t
    fs.close()


def test_of_module_keep_rule_drops_lost_points(pkg, ofk):
    """of_module.py:129 `feasibility-(status-1)>=T` with cv2's uint8 status (:93): status-1 wraps to 255 for a lost point, so it is
    dropped whatever its r, and a tracked point is kept iff r >= T.  With T = -2 every tracked point passes (r is a cosine), hence
    kept == tracked; the clip moves fast enough that border points are lost (rounds 1-2 kept them: r + 1 >= T)."""
    from of_amd import synth
    from of_amd.of_library import pix_trans
    from of_amd.pipeline import FlowStream, PipelineConfig, FusionConfig
    h, w, nf = 240, 320, 5
    cfg = PipelineConfig.of_module()
    cfg.max_corners = 80; cfg.quality = 0.02; cfg.block_size = 7; cfg.min_distance = 8; cfg.max_level = 2; cfg.feas_T = -2.0
    frames, info = synth.render_sequence(h, w, 31, nf, v=(0.05, 0.035, 0.0), omega=(0.0, 0.0, 0.0), d=1.0)     # 16 x 11 px per frame: corners leave the image
    rng = np.random.default_rng(9)
    controls = rng.normal(0, 0.01, (nf - 1, 3)); omegas = rng.normal(0, 0.01, (nf - 1, 3))
    cx, cy = pix_trans((240, 320))
    normal = np.array([0.0, 0.0, 1.0])
    fusion = FusionConfig.of_module(synthetic_flow=False)
    fs = FlowStream(w, h, batch=1, cfg=cfg, min_features=5, mask_radius=10, fusion=fusion)
    tracks, counts = fs.begin(frames[0][None])
    ref = oracle_of_module(frames, cfg, normal, controls, omegas, 5, cx, cy, fusion.model, False)
    lost_total = 0
    for t in range(1, nf):
        sensors = ofk.make_sensors(1, d=1.0, normal=normal, omega=omegas[t - 1], scaling=1.0, cx=cx, cy=cy)
        sensors[:, 25:28] = controls[t - 1]
        rec, fused, tracks, counts = fs.step_fused(frames[t][None], sensors)
        v, xk, P, tr, n_old, n_keep = ref[1][t - 1]
        assert rec[0, 12] == n_old and rec[0, 11] == n_keep and counts[0] == len(tr)
        assert rec[0, 11] == rec[0, 13], "kept == tracked when T is below every cosine"
        assert np.array_equal(tracks[0, :counts[0]].view(np.uint32), tr.astype(np.float32).view(np.uint32))
        lost_total += int(rec[0, 12] - rec[0, 13])
    assert lost_total > 0, "the clip must lose points for this test to mean anything"
    fs.close()


def test_ekf6_with_gps_measurement_block(pkg, ofk, gpu_ctx):
    """FilterModel.ekf6(gps=True): H = [[I 0], [I 0]], R = diag(r I, r_gps I) (nm = 6; DESIGN.md §2a - the reference has no GPS
    data, so this row is build-defined and needs its own known-answer tests):
      1. stand-alone kernel (ofk_kf_predict_update, nm = 6) == the numpy restatement;
      2. information-fusion identity: two independent measurements z1 (variance r) and z2 (variance r_gps) of the same state are
         one measurement (z1 / r + z2 / r_gps) / (1 / r + 1 / r_gps) with variance 1 / (1 / r + 1 / r_gps) - the 6-row update must
         land on the 3-row update with those, to rounding;
      3. through the resident stream loop (k_stream_fuse reads the second block from sensor slots 22-24) == the oracle
         composition of the node loop with the stacked measurement, and == the stand-alone kernel fed the same numbers."""
    from of_amd import synth
    from of_amd.pipeline import FlowStream, PipelineConfig, FusionConfig, FilterModel
    from oracle import estimation_oracle as eo
    m6 = FilterModel.ekf6(dt=0.1, r=4.0, gps=True, r_gps=0.5)
    m3 = FilterModel.ekf6(dt=0.1, r=1.0 / (1.0 / 4.0 + 1.0 / 0.5), gps=False)
    assert m6.nm == 6 and m6.H.shape == (6, 6) and m6.R.shape == (6, 6)
    rng = np.random.default_rng(21)
    x6, P6 = m6.x0.copy(), m6.P0.copy(); x3, P3 = m3.x0.copy(), m3.P0.copy(); xn, Pn = m6.x0.copy(), m6.P0.copy()
    for t in range(6):
        u = rng.normal(0, 0.05, 3); z1 = rng.normal(0.3, 0.2, 3); z2 = z1 + rng.normal(0, 0.05, 3)
        x6, P6 = gpu_ctx.kf_predict_update(m6.F, m6.H, m6.Q, m6.R, x6, P6, B=m6.B, u=u, z=np.concatenate([z1, z2]))
        xn, Pn = eo.kf_predict(xn, Pn, m6.F, m6.Q, m6.B, u); xn, Pn = eo.kf_correct(xn, Pn, m6.H, m6.R, np.concatenate([z1, z2]))
        zeff = (z1 / 4.0 + z2 / 0.5) / (1.0 / 4.0 + 1.0 / 0.5)
        x3, P3 = gpu_ctx.kf_predict_update(m3.F, m3.H, m3.Q, m3.R, x3, P3, B=m3.B, u=u, z=zeff)
        np.testing.assert_allclose(x6, xn, rtol=1e-11, atol=1e-14); np.testing.assert_allclose(P6, Pn, rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(x6, x3, rtol=1e-10, atol=1e-13); np.testing.assert_allclose(P6, P3, rtol=1e-9, atol=1e-13)
    # 3. the resident loop
    h, w, nf = 240, 320, 6
    cfg = PipelineConfig(max_corners=80, quality=0.02, min_distance=8, block_size=7, max_level=2)
    frames, info = synth.render_sequence(h, w, 951, nf, v=(0.004, -0.003, 0.002), omega=(0.002, 0.001, -0.003), d=1.0)
    statics = dict(d=1.0, offset=(0.0, 0.0, 0.1), scaling=info["scaling"], cx=info["cx"], cy=info["cy"])
    msgs = np.stack([make_imu_msgs(rng, 50.0 + 0.1 * t, 3) for t in range(nf - 1)])          # [nf-1, 3, 15]
    gps = rng.normal(0.0, 0.01, (nf - 1, 3)) + [0.004, -0.003, 0.002]
    fusion = FusionConfig.ekf6(dt=0.1, gps=True, r=4.0, r_gps=0.5)
    fs = FlowStream(w, h, batch=1, cfg=cfg, min_features=70, mask_radius=12, fusion=fusion)
    fs.begin(frames[0][None])
    ref = oracle_node_fused(frames, cfg, statics, msgs, 70, 12, fusion.model, gps=gps)
    xs, Ps = fusion.model.x0.copy(), fusion.model.P0.copy()
    for t in range(1, nf):
        fs.push_imu(msgs[t - 1][None])
        _, dv = fs.ctx.imu_state(1)                              # the increments the step will consume as the filter's control
        sensors = ofk.make_sensors(1, d=statics["d"], offset=statics["offset"], scaling=statics["scaling"], cx=statics["cx"], cy=statics["cy"],
                                   v_prior=gps[t - 1])           # slots 22-24: the second measurement block
        rec, fused, tracks, counts = fs.step_fused(frames[t][None], sensors)
        v, vu, xk, tr, n_old, n_tr = ref[1][t - 1]
        assert rec[0, 15] == 1 and counts[0] == len(tr)
        np.testing.assert_allclose(rec[0, 8:11], vu, rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(fused[0, :6], xk, rtol=1e-8, atol=1e-12)
        xs, Ps = gpu_ctx.kf_predict_update(fusion.model.F, fusion.model.H, fusion.model.Q, fusion.model.R, xs, Ps, B=fusion.model.B, u=dv[0],
                                           z=np.concatenate([rec[0, 8:11], gps[t - 1]]))
        gx, gP = fs.ctx.filter_state(1)
        assert np.array_equal(gx[0].view(np.uint64), xs.view(np.uint64)) and np.array_equal(gP[0].view(np.uint64), Ps.view(np.uint64)), t
    fs.close()


@pytest.mark.parametrize("T", [0.2, 0.8])
def test_of_module_continue_keeps_the_old_frame(pkg, ofk, T):
    """of_module.py:138: with <= 3 feasible points the script `continue`s - old_gray and old_pos stay, the next frame is tracked from
    the OLD one, the filter keeps its prediction.  FusionConfig.of_module(hold_on_skip=True) on a FlowStream of one stream against
    the oracle loop with the same rule; both thresholds give clips with held AND solved steps.  A batch of streams refuses the flag."""
    from of_amd import synth
    from of_amd.of_library import pix_trans
    from of_amd.pipeline import FlowStream, PipelineConfig, FusionConfig
    h, w, nf = 240, 320, 9
    frames, info = synth.render_sequence(h, w, 41, nf, v=(0.006, -0.004, 0.002), omega=(0.001, -0.002, 0.003), d=1.0)
    rng = np.random.default_rng(13)
    controls = rng.normal(0, 0.01, (nf - 1, 3)); omegas = rng.normal(0, 0.01, (nf - 1, 3))
    cx, cy = pix_trans((240, 320))
    normal = np.array([0.0, 0.0, 1.0])
    cfg = PipelineConfig.of_module(); cfg.max_corners = 60; cfg.quality = 0.05; cfg.block_size = 7; cfg.min_distance = 8; cfg.max_level = 2; cfg.feas_T = T
    fusion = FusionConfig.of_module(synthetic_flow=False, hold_on_skip=True)
    ref = oracle_of_module(frames, cfg, normal, controls, omegas, 10, cx, cy, fusion.model, False, hold=True)
    fs = FlowStream(w, h, batch=1, cfg=cfg, min_features=10, mask_radius=10, fusion=fusion)
    tracks, counts = fs.begin(frames[0][None])
    held = solved = 0
    for t in range(1, nf):
        sensors = ofk.make_sensors(1, d=1.0, normal=normal, omega=omegas[t - 1], scaling=1.0, cx=cx, cy=cy)
        sensors[:, 25:28] = controls[t - 1]
        rec, fused, tracks, counts = fs.step_fused(frames[t][None], sensors)
        v, xk, P, tr, n_old, n_keep = ref[1][t - 1]
        assert rec[0, 12] == n_old and rec[0, 11] == n_keep and counts[0] == len(tr), (t, rec[0, 11:14], n_old, n_keep, len(tr))
        assert np.array_equal(tracks[0, :counts[0]].view(np.uint32), tr.astype(np.float32).view(np.uint32)), t
        np.testing.assert_allclose(fused[0, :3], xk, rtol=1e-8, atol=1e-12)
        if v is None:
            held += 1
            assert rec[0, 15] == 0
        else:
            solved += 1
            np.testing.assert_allclose(rec[0, :3], v, rtol=1e-7, atol=1e-12)
    assert held >= 1 and solved >= 2
    fs.close()
    fs2 = FlowStream(w, h, batch=2, cfg=cfg, min_features=10, mask_radius=10, fusion=fusion)
    fs2.begin(np.stack([frames[0], frames[0]]))
    with pytest.raises(ofk.OfkError):
        fs2.step_fused(np.stack([frames[1], frames[1]]), ofk.make_sensors(2, scaling=1.0, cx=cx, cy=cy))
    fs2.close()
