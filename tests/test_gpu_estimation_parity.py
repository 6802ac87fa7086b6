"""GPU parity, estimation stages (float64): HIP kernels through the C ABI vs the golden vectors produced by the
reference's own numpy functions and vs oracle/estimation_oracle.py.  Tolerance: 1e-10 relative on velocities and
singular values (north_star asks 1e-4), residual sums of squares 1e-8 relative + 1e-20 absolute."""
import numpy as np
import pytest

from oracle import estimation_oracle as eo

pytestmark = pytest.mark.gpu
RT = 1e-10


def close(a, b, rtol=RT, atol=1e-13):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


def test_flow_model_golden(gpu_ctx, golden):
    g = golden
    close(gpu_ctx.flow_model(g["g1_points"], g["g1_v"], g["g1_omega"], float(g["g1_d"]), g["g1_n"], g["g1_t"]), g["g1_flow"])
    close(gpu_ctx.flow_model(g["g2_x"], [1, 1, 1], [0, 0, 0], 0.75, [0, 0, 1]), g["g2_u"])
    # batched, per-problem parameters
    rng = np.random.default_rng(0)
    x = rng.uniform(-.5, .5, (5, 37, 2)); v = rng.normal(0, 1, (5, 3)); om = rng.normal(0, .2, (5, 3)); d = rng.uniform(.5, 3, 5)
    n = rng.normal(0, .1, (5, 3)) + [0, 0, 1]; t = rng.normal(0, .1, (5, 3))
    got = gpu_ctx.flow_model(x, v, om, d, n, t)
    for b in range(5):
        close(got[b], eo.generate_test_data(x[b], v[b], om[b], d[b], n[b], t[b]))


@pytest.mark.parametrize("N", [3, 4, 20, 200, 500, 2000])
def test_solve_golden_all_variants(gpu_ctx, ofk, golden, N):
    g = golden; p = f"g4_{N}_"
    x, u, n, om, d, t = g[p + "x"], g[p + "u"], g[p + "n"], g[p + "omega"], float(g[p + "d"]), g[p + "t"]
    o = gpu_ctx.velocity_solve(ofk.SOLVE_NODE, x, u, d=d, nrm=n, omega=om)
    close(o[:3], g[p + "node_v"]); close(o[5:8], g[p + "node_s"]); assert o[4] == g[p + "node_rank"]
    if N > 1 and len(g[p + "node_R"]):
        close(o[3], g[p + "node_R"][0], rtol=1e-8, atol=1e-20)
    o = gpu_ctx.velocity_solve(ofk.SOLVE_SIM, x, u, d=d, nrm=n, omega=om, t=t)
    close(o[:3], g[p + "sim_v"]); close(o[5:8], g[p + "sim_s"])
    if len(g[p + "sim_R"]):
        close(o[3], g[p + "sim_R"][0], rtol=1e-8, atol=1e-20)
    o = gpu_ctx.velocity_solve(ofk.SOLVE_NODE, x, u, d=d, nrm=n, omega=om, t=t)
    close(o[:3], g[p + "eval_v"])


def test_solve_kats_rank_and_valid(gpu_ctx, ofk, golden):
    g = golden
    o = gpu_ctx.velocity_solve(ofk.SOLVE_NODE, g["g2_x"], g["g2_u"], d=0.75, nrm=[0, 0, 1], omega=[0, 0, 0])
    close(o[:3], [1, 1, 1]); close(o[5:8], g["g2_s"]); assert o[4] == 3 and o[3] < 1e-24
    o = gpu_ctx.velocity_solve(ofk.SOLVE_SIM, g["g1_points"], g["g1_flow"], d=1.0, nrm=g["g1_n"], omega=g["g1_omega"], t=g["g1_t"])
    close(o[:3], [1, 1, 1]); close(o[5:8], g["g1_s"])
    # rank-deficient system: minimum-norm solution like lstsq, rank 2
    o = gpu_ctx.velocity_solve(ofk.SOLVE_NODE, g["g4_def_x"], g["g4_def_u"], d=1.3, nrm=[0, 0, 1.0], omega=[0.1, 0.0, -0.2])
    assert o[4] == 2 == int(g["g4_def_rank"]); close(o[:3], g["g4_def_v"]); close(o[5:7], g["g4_def_s"][:2])
    # of_module system
    o = gpu_ctx.velocity_solve(ofk.SOLVE_OFMODULE, g["g9_x"][:, :2], g["g9_u"][:, :2], nrm=[0, 0, 1], wgt=g["g9_dist"])
    close(o[:3], g["g9_v"]); close(o[5:8], g["g9_s"]); close(o[3], g["g9_R"][0], rtol=1e-8)
    # valid mask == solving the subset; batch of problems
    x, u = g["g4_200_x"], g["g4_200_u"]
    valid = (np.arange(200) % 3 != 0).astype(np.uint8)
    a = gpu_ctx.velocity_solve(ofk.SOLVE_NODE, np.stack([x, x]), np.stack([u, u]), d=[2.0, 2.0], nrm=g["g4_200_n"],
                               omega=g["g4_200_omega"], valid=np.stack([valid, np.ones(200, np.uint8)]))
    b = gpu_ctx.velocity_solve(ofk.SOLVE_NODE, x[valid == 1], u[valid == 1], d=2.0, nrm=g["g4_200_n"], omega=g["g4_200_omega"])
    close(a[0], b, rtol=1e-11)
    v, R, rank, s = eo.solve_lgs_node(x, u, 2.0, g["g4_200_n"], g["g4_200_omega"])
    close(a[1, :3], v); close(a[1, 5:8], s)
    # no valid point -> rank 0, zeros
    z = gpu_ctx.velocity_solve(ofk.SOLVE_NODE, x, u, d=2.0, nrm=[0, 0, 1], omega=[0, 0, 0], valid=np.zeros(200, np.uint8))
    assert z[4] == 0 and np.all(z[:3] == 0)


def test_feasibility_golden(gpu_ctx, ofk, golden):
    g = golden
    r, d = gpu_ctx.feasibility(ofk.FEAS_RTILDE, g["g2_x"], g["g2_u"], [0, 0, 1], [.1, .1, .1], dist=.75)
    close(r, -np.ones(4)); close(d, 0.1 * np.ones(4))
    r, d = gpu_ctx.feasibility(ofk.FEAS_RTILDE, g["g3r_x"], g["g3r_u"], g["g3r_n"], g["g3r_v"], dist=float(g["g3r_dist"]))
    close(r, g["g3r_r"]); close(d, g["g3r_d"]); assert r[5] == 1.0 and d[5] == 1.0
    r, d = gpu_ctx.feasibility(ofk.FEAS_RTILDE, g["g3r_x"], g["g3r_u"], -g["g3r_n"], g["g3r_v"], dist=float(g["g3r_dist"]))
    close(r, g["g3n_r"]); close(d, g["g3n_d"])
    keep = g["g3c_keep"]
    r, d = gpu_ctx.feasibility(ofk.FEAS_LEGACY, g["g3r_x"][keep], g["g3r_u"][keep], g["g3r_n"], g["g3r_v"])
    close(r, g["g3c_r"]); close(d, g["g3c_d"])
    r, d = gpu_ctx.feasibility(ofk.FEAS_SIM, g["g1_points"], g["g1_flow"], g["g1_n"], g["g1_v"], omega=g["g1_omega"], t=g["g1_t"])
    close(np.array([r, d]), g["g3b_out"])


def test_imu_and_post_solve_golden(gpu_ctx, ofk, golden):
    g = golden
    st = np.zeros(ofk.IMU_STATE); st[0:3] = g["g5_vel0"]; st[5] = 1.0
    for m, want in zip(g["g5_msgs"], g["g5_states"]):
        msg = np.concatenate([m[0:9], g["g5_cov_diag"], m[9:12]])
        st = gpu_ctx.imu_propagate(st, msg)
        got = np.concatenate([st[0:5], st[6:24]])
        close(got, want, rtol=1e-13, atol=1e-15)
    close(gpu_ctx.post_solve(g["g10_v_obs"], g["g10_rotation"], g["g10_ang"], g["g10_offset"]), g["g10_v_uav"], rtol=1e-14)


def test_kf_reference_matrices_and_6_state(gpu_ctx):
    # of_module.py:63-76: F=B=H=I, Q=1e-5 I, R=10 I, P0=.1 I, x0=0 -> scalar recursion per axis (closed-form KAT)
    I = np.eye(3)
    rng = np.random.default_rng(1)
    B = 7
    x = np.zeros((B, 3)); P = np.tile(0.1 * I, (B, 1, 1))
    xs = np.zeros((B, 3)); ps = 0.1
    for _ in range(20):
        u = rng.normal(0, 0.01, (B, 3)); z = rng.normal(0, 1, (B, 3))
        x, P = gpu_ctx.kf_predict_update(I, I, 1e-5 * I, 10 * I, x, P, B=I, u=u, z=z)
        xs = xs + u; ps = ps + 1e-5
        k = ps / (ps + 10); xs = xs + k * (z - xs); ps = (1 - k) * ps
        close(x, xs, rtol=1e-13); close(P, np.tile(ps * I, (B, 1, 1)), rtol=1e-13, atol=1e-18)
    # generic 6-state / 3-measurement filter against the numpy restatement
    ns, nm = 6, 3
    F = np.eye(ns) + 0.05 * rng.normal(size=(ns, ns)); H = rng.normal(size=(nm, ns)); Q = 0.01 * np.eye(ns)
    A = rng.normal(size=(nm, nm)); R = A @ A.T + np.eye(nm); Bm = rng.normal(size=(ns, 3))
    x = rng.normal(size=(4, ns)); P0 = rng.normal(size=(ns, ns)); P = np.tile(P0 @ P0.T + np.eye(ns), (4, 1, 1))
    u = rng.normal(size=(4, 3)); z = rng.normal(size=(4, nm))
    gx, gP = gpu_ctx.kf_predict_update(F, H, Q, R, x, P, B=Bm, u=u, z=z)
    for b in range(4):
        xr, Pr = eo.kf_predict(x[b], P[b], F, Q, Bm, u[b]); xr, Pr = eo.kf_correct(xr, Pr, H, R, z[b])
        close(gx[b], xr, rtol=1e-11); close(gP[b], Pr, rtol=1e-10, atol=1e-12)
    # predict only / correct only split like cv2.KalmanFilter.predict() then .correct()
    px, pP = gpu_ctx.kf_predict_update(F, H, Q, R, x, P, B=Bm, u=u, z=None)
    cx, cP = gpu_ctx.kf_predict_update(F, H, Q, R, px, pP, z=z, do_predict=False)
    close(cx, gx, rtol=1e-13); close(cP, gP, rtol=1e-13)


@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_of_simulation_golden(gpu_ctx, golden, lvl):
    g = golden
    truth = np.concatenate([g["g1_v"], g["g1_omega"], [1.0], g["g1_n"], g["g1_t"]])
    v, bound = gpu_ctx.of_simulation(truth, g[f"g6_{lvl}_sig"], g["g1_points"], g["g1_flow"], g[f"g6_{lvl}_z"])
    close(v, g[f"g6_{lvl}_v_obs"]); close(bound, g[f"g6_{lvl}_bound"], rtol=1e-8)


def test_of_simulation_statistics_vs_saved_sweep(gpu_ctx, golden):
    """Statistical pin (unseeded RNG in the reference): sigma-step 40 of effect_of_flow_errors.npy
    (simulation.py:183-202) — mean within 4 sigma/sqrt(n), std within 35 %."""
    g = golden
    saved = g["g8_effect_of_flow_errors"]; mean_s, std_s = saved[:300].reshape(100, 3), saved[300:].reshape(100, 3)
    data = g["g1_points"].copy()
    data[:, 0] = (data[:, 0] - data[:, 0].mean()) * 1.27; data[:, 1] = (data[:, 1] - data[:, 1].mean()) * 0.93
    tf = eo.generate_test_data(data, g["g1_v"], g["g1_omega"], 1.0, g["g1_n"], g["g1_t"])
    truth = np.concatenate([g["g1_v"], g["g1_omega"], [1.0], g["g1_n"], g["g1_t"]])
    i = 40
    sig = [0.00071, 0.005, 0.01, 0.001 * i, np.sqrt(2) / 1000 * i, 0.00065]
    z = np.random.default_rng(5).standard_normal((2000, 10 + 4 * 200))
    v, _ = gpu_ctx.of_simulation(truth, sig, data, tf, z)
    assert np.all(np.abs(v.mean(0) - mean_s[i]) < 4 * std_s[i] / np.sqrt(100) + 4 * v.std(0) / np.sqrt(2000))
    assert np.all(np.abs(v.std(0) / std_s[i] - 1) < 0.35)


# ---- the eight saved Monte-Carlo sweeps (simulation.py:183-461): every effect_*.npy the reference holds is a statistical pin
SWEEP_RULES = {
    # axis: (sigma overrides, n_ref = the block's `iterations`, std band (None: not compared), steps whose std is compared,
    #        extra tolerance on the means, components compared)
    # --- curves the script AS COMMITTED reproduces: mean within 4 sigma / sqrt(n) of both samples, std within the band
    "flow_errors":       ({}, 100, (0.65, 1.35), (10, 50, 90), 0.0, (0, 1, 2)),
    "point_position":    ({}, 100, (0.65, 1.35), (10, 50, 90), 0.0, (0, 1, 2)),
    "orientation":       ({}, 10, (0.4, 2.5), (10, 50, 90), 0.0, (0, 1, 2)),       # 10 trials per step in the reference: its std estimate scatters by 24 %
    # --- effect_of_distance_error.npy predates the "* 1.23" the author later put on flow_sig / position_sig (:169-170): with the
    #     constants without it the curve is reproduced as tightly as the three above (v_z bias 0.971 instead of 0.956)
    "distance_error":    ({"flow_sig": 0.056 * np.sqrt(2), "position_sig": 0.056}, 100, (0.65, 1.35), (10, 50, 90), 0.0, (0, 1, 2)),
    # --- curves saved under base constants the script no longer holds (v_z bias 0.98-0.99 instead of 0.956: smaller position noise at
    #     the time): the swept sigma dominates the spread from the middle of the sweep on, which is what is compared; means get
    #     the bias difference as tolerance
    "ang_vel_error":     ({}, 100, (0.65, 1.35), (50, 90), 0.045, (0, 1)),
    "translation_error": ({}, 100, (0.65, 1.35), (50, 90), 0.045, (0, 1, 2)),
    "normal_error":      ({}, 10, (0.3, 3.0), (10, 50, 90), 0.045, (0, 1, 2)),       # the normal draw is discarded (:45-46): a flat curve, 10 trials per step
    # --- effect_of_height.npy: x / y spread 3-5 x what the committed constants give (the block carries commented-out alternative
    #     velocities, :408-409): only the means are comparable
    "height":            ({}, 100, None, (), 0.06, (0, 1, 2)),
}


@pytest.fixture(scope="module")
def sweeps_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_sweeps.npz"))


@pytest.mark.parametrize("axis", sorted(SWEEP_RULES))
def test_saved_sweep_is_reproduced(pkg, gpu_ctx, sweeps_golden, axis):
    """simulation.sweep(axis) - one launch of k_of_simulation per step, 2000 trials - against the curve the reference saved for
    that block, at steps 10, 50 and 90 of 100 (unseeded RNG in the reference: a statistical comparison)."""
    import of_amd.simulation as sim
    override, n_ref, band, std_steps, mean_tol, comps = SWEEP_RULES[axis]
    saved = sweeps_golden[sim.SWEEP_AXES[axis][0]]
    mean_s, std_s = saved[:300].reshape(100, 3), saved[300:].reshape(100, 3)
    steps, trials = (10, 50, 90), 2000
    out = sim.sweep(axis, sweeps_golden["points_raw"], [1.0, 1, 1], [1.0, 1, 1], 1.0, [0.0, 0, 1], [0.02, 0, 0.205], sigmas=override,
                    k=100, trials=trials, steps=steps, generator=np.random.default_rng(11))
    mean_o, std_o = out[:9].reshape(3, 3), out[9:].reshape(3, 3)
    assert np.all(np.isfinite(out))
    for row, i in enumerate(steps):
        for c in comps:
            tol = 4 * std_s[i, c] / np.sqrt(n_ref) + 4 * std_o[row, c] / np.sqrt(trials) + mean_tol
            assert abs(mean_o[row, c] - mean_s[i, c]) < tol, (axis, i, c, mean_o[row], mean_s[i], tol)
            if band is not None and i in std_steps:
                assert band[0] < std_o[row, c] / std_s[i, c] < band[1], (axis, i, c, std_o[row], std_s[i])


def test_sweep_layout_and_default_trials(pkg, gpu_ctx, sweeps_golden):
    """Full-length call: np.append(v_mean, v_std) of k x 3 + k x 3 values like the saved files; `trials` defaults to the block's own
    `iterations`; the module's `iterations` global is left as it was."""
    import of_amd.simulation as sim
    before = sim.iterations
    out = sim.sweep("orientation", sweeps_golden["points_raw"], [1.0, 1, 1], [1.0, 1, 1], 1.0, [0.0, 0, 1], [0.02, 0, 0.205], k=12,
                    generator=np.random.default_rng(2))
    assert out.shape == (72,) and np.all(np.isfinite(out)) and sim.iterations == before
    with pytest.raises(ValueError):
        sim.sweep_step("no_such_axis", 0, 100, sweeps_golden["points_raw"], 1.0, [0, 0, 1], None)


# ---- the counter-based noise of the device (ofk_of_simulation_rng): generator, kernel and sweep against the oracle's restatement
def test_device_noise_generator_matches_the_oracle(gpu_ctx):
    """ofk_noise_normals = oracle noise_normals: Philox4x32-10 integers identical, the f64 Box-Muller transform to the last ulps of
    libm's log / sin / cos (a normal near zero is a difference of magnitudes ~1: absolute tolerance 4e-16)."""
    for seed, step, trial, count in ((0, 0, 0, 8), (1, 2, 3, 811), (0xFFFFFFFFFFFFFFFF, 99, 4095, 8010), (20261005, 7, 123456, 33)):
        got = gpu_ctx.noise_normals(seed, step, trial, count)
        ref = eo.noise_normals(seed, step, trial, count)
        np.testing.assert_allclose(got, ref, rtol=2e-14, atol=4e-16)
    z = gpu_ctx.noise_normals(5, 1, 2, 400000)
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.std() - 1) < 5e-3 and abs((z ** 4).mean() - 3) < 0.05


@pytest.mark.parametrize("n_pts,trials", [(200, 64), (2000, 24)])
def test_of_simulation_with_device_noise_equals_injected_oracle_noise(gpu_ctx, golden, n_pts, trials):
    """The kernel with RNG = true draws exactly the rows the oracle generates: v_obs and the analytic bound equal the injected-noise
    kernel's (golden G6 pins that one to the reference) on those rows; a shard (trial0 > 0) equals the slice of the whole."""
    g = golden
    rng = np.random.default_rng(n_pts)
    pts = g["g1_points"] if n_pts == 200 else rng.uniform(-1.2, 1.2, (n_pts, 2))
    tf = eo.generate_test_data(pts, g["g1_v"], g["g1_omega"], 1.0, g["g1_n"], g["g1_t"])
    truth = np.concatenate([g["g1_v"], g["g1_omega"], [1.0], g["g1_n"], g["g1_t"]])
    sig = [0.00071, 0.005, 0.01, 0.02, 0.03, 0.00065]
    seed, step = 0x1234ABCD5678, 17
    v, b = gpu_ctx.of_simulation_rng(truth, sig, pts, tf, seed, step, trials)
    z = eo.noise_rows(seed, step, 0, trials, n_pts)
    v_ref, b_ref = gpu_ctx.of_simulation(truth, sig, pts, tf, z)
    close(v, v_ref, rtol=1e-9); close(b, b_ref, rtol=1e-8)
    v_sh, b_sh = gpu_ctx.of_simulation_rng(truth, sig, pts, tf, seed, step, trials // 2, trial0=trials // 4)
    assert np.array_equal(v_sh, v[trials // 4:trials // 4 + trials // 2]) and np.array_equal(b_sh, b[trials // 4:trials // 4 + trials // 2])
    v_other, _ = gpu_ctx.of_simulation_rng(truth, sig, pts, tf, seed, step + 1, trials)
    assert not np.allclose(v_other, v)


def test_sweep_with_device_noise_reproduces_a_saved_curve(pkg, gpu_ctx, sweeps_golden):
    """simulation.sweep(device_seed=...) at the batch width of BASELINE configs[4] (4096 trials per step, nothing uploaded but the
    points): statistics against the reference's saved flow-error curve under the same rule as the injected-noise sweep."""
    import of_amd.simulation as sim
    axis = "flow_errors"
    override, n_ref, band, std_steps, mean_tol, comps = SWEEP_RULES[axis]
    saved = sweeps_golden[sim.SWEEP_AXES[axis][0]]
    mean_s, std_s = saved[:300].reshape(100, 3), saved[300:].reshape(100, 3)
    steps, trials = (10, 50, 90), 4096
    out = sim.sweep(axis, sweeps_golden["points_raw"], [1.0, 1, 1], [1.0, 1, 1], 1.0, [0.0, 0, 1], [0.02, 0, 0.205], sigmas=override,
                    k=100, trials=trials, steps=steps, device_seed=424242)
    mean_o, std_o = out[:9].reshape(3, 3), out[9:].reshape(3, 3)
    for row, i in enumerate(steps):
        for c in comps:
            tol = 4 * std_s[i, c] / np.sqrt(n_ref) + 4 * std_o[row, c] / np.sqrt(trials) + mean_tol
            assert abs(mean_o[row, c] - mean_s[i, c]) < tol, (i, c, mean_o[row], mean_s[i], tol)
            if band is not None and i in std_steps:
                assert band[0] < std_o[row, c] / std_s[i, c] < band[1], (i, c, std_o[row], std_s[i])
    again = sim.sweep(axis, sweeps_golden["points_raw"], [1.0, 1, 1], [1.0, 1, 1], 1.0, [0.0, 0, 1], [0.02, 0, 0.205], sigmas=override,
                      k=100, trials=trials, steps=steps, device_seed=424242)
    assert np.array_equal(again, out)                            # a seed names a sweep


# ---- feas_simulation + overlap (simulation.py:70-104, 124-136; the live experiment :753-774), golden from the reference's functions
@pytest.fixture(scope="module")
def feas_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_feas.npz"))


@pytest.mark.parametrize("case", [0, 1])
def test_feas_simulation_matches_reference(gpu_ctx, feas_golden, case):
    g = feas_golden
    z = g[f"f{case}_z"]
    mean, v_obs, per = gpu_ctx.feas_simulation(g[f"f{case}_truth"], g[f"f{case}_sig"], g[f"f{case}_pos"], g[f"f{case}_true_flow"], z, per_trial=True)
    np.testing.assert_allclose(mean, g[f"f{case}_mean"], rtol=1e-9, atol=1e-12)
    tr = g[f"f{case}_truth"]
    om, vo, tab = eo.feas_simulation(tr[3:6], tr[6], tr[7:10], tr[10:13], g[f"f{case}_pos"], g[f"f{case}_true_flow"], tr[13:16], g[f"f{case}_sig"], z,
                                     len(z), per_trial=True)
    np.testing.assert_allclose(v_obs, vo, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(per, tab, rtol=1e-9, atol=1e-12)
    a, b = g[f"f{case}_split"]
    for q in range(6):                                          # the three planes' histograms of every statistic (:788-810)
        r = g[f"f{case}_mean"][q]
        got = [gpu_ctx.hist_overlap(r[:a], r[a:b]), gpu_ctx.hist_overlap(r[a:b], r[b:]), gpu_ctx.hist_overlap(r[:a], r[b:])]
        assert got == list(g[f"f{case}_overlap"][q]), (q, got)


def test_hist_overlap_edges_and_degenerate_samples(gpu_ctx, feas_golden):
    g = feas_golden
    assert gpu_ctx.hist_overlap(g["ov_d1"], g["ov_d2"]) == int(g["ov_12"])
    assert gpu_ctx.hist_overlap(g["ov_d3"], g["ov_d3"]) == int(g["ov_33"]) == 40     # all values equal: numpy widens the range by 0.5
    assert gpu_ctx.hist_overlap(g["ov_d1"], g["ov_d3"]) == int(g["ov_13"])
    rng = np.random.default_rng(5)
    for n1, n2, bins in ((1, 1, 100), (1000, 3, 7), (257, 4096, 1024), (50, 50, 1)):
        a, b = rng.normal(0, 1, n1), rng.normal(0.3, 2, n2)
        assert gpu_ctx.hist_overlap(a, b, bins) == eo.overlap(a, b, bins), (n1, n2, bins)
    with pytest.raises(Exception):
        gpu_ctx.hist_overlap(a, b, 2000)


def test_simulation_facade_feas_simulation_and_sweep_statistics(pkg, feas_golden):
    """Drop-in call with the reference's positional signature (module globals true_flow / iterations / normal_sig / velocity_sig),
    noise injected; then the experiment's read-out (:778-779): the forward parallelity separates the rotated-flow plane."""
    from of_amd import simulation as sim
    g = feas_golden
    tr, sg = g["f0_truth"], g["f0_sig"]
    sim.true_flow = g["f0_true_flow"]; sim.iterations = len(g["f0_z"]); sim.normal_sig = float(sg[5]); sim.velocity_sig = float(sg[6])
    out = sim.feas_simulation(tr[0:3], tr[3:6], tr[6], tr[7:10], tr[10:13], g["f0_pos"], sg[0], sg[1], sg[2], sg[3], sg[4], sg[5], tr[13:16], z=g["f0_z"])
    assert len(out) == 6
    np.testing.assert_allclose(np.stack(out), g["f0_mean"], rtol=1e-9, atol=1e-12)
    a, b = g["f0_split"]
    assert sim.overlap(out[2][:a], out[2][a:b]) == int(g["f0_overlap"][2][0])
    # static planes have forward parallelity near +1, the plane with randomly rotated flow does not
    assert np.median(out[2][:a]) > 0.9 and np.median(out[2][b:]) > 0.9 and np.mean(out[2][a:b] > 0.88) < 0.35
