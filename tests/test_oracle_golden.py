"""Pins oracle/estimation_oracle.py against golden vectors produced by the reference's own numpy
functions (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import estimation_oracle as eo

TOL = 1e-12


def close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


def test_g1_roundtrip_points_txt(golden):
    g = golden
    flow = eo.generate_test_data(g["g1_points"], g["g1_v"], g["g1_omega"], float(g["g1_d"]), g["g1_n"], g["g1_t"])
    close(flow, g["g1_flow"])
    assert abs(flow[0, 0] - 2.593068975207) < 1e-11 and abs(flow[0, 1] - 0.286187463185) < 1e-11
    v, R, s = eo.solve_lgs_sim(g["g1_points"], flow, float(g["g1_d"]), g["g1_n"], g["g1_omega"], g["g1_t"])
    close(v, g["g1_v_out"]); close(v, [1, 1, 1]); close(s, g["g1_s"])
    assert R.shape == (1,) and R[0] < 1e-24
    vn, Rn, rank, sn = eo.solve_lgs_node(g["g1_points"], flow, float(g["g1_d"]), g["g1_n"], g["g1_omega"])
    close(vn, g["g1_node_v"]); close(vn, [1.205, 0.815, 0.98]); close(sn, g["g1_node_s"]); assert rank == 3


def test_g2_node_kat(golden):
    g = golden
    tr = eo.pix_trans((320, 240))
    x = g["g2_feat"].astype(float)
    x[:, 0] = (x[:, 0] - tr[0]) * 0.01; x[:, 1] = (x[:, 1] - tr[1]) * 0.01
    close(x, g["g2_x"])
    u = eo.generate_test_data(x, [1, 1, 1], [0, 0, 0], 0.75, [0, 0, 1])
    close(u, g["g2_u"])
    v, R, rank, s = eo.solve_lgs_node(x, u, 0.75, [0, 0, 1], [0, 0, 0])
    close(v, [1, 1, 1]); close(s, g["g2_s"]); assert rank == int(g["g2_rank"]) == 3


def test_g3_feasibility(golden):
    g = golden
    r, d = eo.r_tilde(g["g2_x"], g["g2_u"], [0, 0, 1], [.1, .1, .1], .75)
    close(r, g["g3_r"]); close(d, g["g3_d"]); close(r, -np.ones(4)); close(d, 0.1 * np.ones(4))
    r, d = eo.r_tilde(g["g3r_x"], g["g3r_u"], g["g3r_n"], g["g3r_v"], float(g["g3r_dist"]))
    close(r, g["g3r_r"]); close(d, g["g3r_d"])
    assert r[5] == 1.0 and d[5] == 1.0          # zero-norm guard
    r, d = eo.r_tilde(g["g3r_x"], g["g3r_u"], -g["g3r_n"], g["g3r_v"], float(g["g3r_dist"]))
    close(r, g["g3n_r"]); close(d, g["g3n_d"])
    fe = eo.feasibility_sim(g["g1_points"], g["g1_v"], g["g1_flow"], g["g1_omega"], g["g1_t"], g["g1_n"])
    close(fe, g["g3b_out"])
    close(fe[0, :3], [0.952212343398, 0.850736111959, 0.951098920789], 1e-11)
    keep = g["g3c_keep"]
    x3 = np.concatenate([g["g3r_x"], np.ones((64, 1))], 1)[keep]
    u3 = np.concatenate([g["g3r_u"], np.zeros((64, 1))], 1)[keep]
    r, d = eo.r_tilde_legacy(x3, u3, g["g3r_n"], g["g3r_v"])
    close(r, g["g3c_r"]); close(d, g["g3c_d"])


@pytest.mark.parametrize("N", [3, 4, 20, 200, 500, 2000])
def test_g4_solve_variants(golden, N):
    g = golden; p = f"g4_{N}_"
    x, u, n, om, d, t = g[p + "x"], g[p + "u"], g[p + "n"], g[p + "omega"], float(g[p + "d"]), g[p + "t"]
    v, R, rank, s = eo.solve_lgs_node(x, u, d, n, om)
    close(v, g[p + "node_v"], 1e-10); close(R, g[p + "node_R"], 1e-9); close(s, g[p + "node_s"], 1e-11)
    assert rank == int(g[p + "node_rank"])
    v, R, s = eo.solve_lgs_sim(x, u, d, n, om, t)
    close(v, g[p + "sim_v"], 1e-10); close(R, g[p + "sim_R"], 1e-9); close(s, g[p + "sim_s"], 1e-11)
    v, R = eo.solve_lgs_eval(x, u, d, n, om, t)
    close(v, g[p + "eval_v"], 1e-10); close(R, g[p + "eval_R"], 1e-9)


def test_g4_rank_deficient(golden):
    g = golden
    v, R, rank, s = eo.solve_lgs_node(g["g4_def_x"], g["g4_def_u"], 1.3, [0, 0, 1.0], [0.1, 0.0, -0.2])
    assert rank == 2 == int(g["g4_def_rank"]) and R.shape == (0,)
    close(v, g["g4_def_v"], 1e-10); close(s[:2], g["g4_def_s"][:2])


def test_g5_call_imu(golden):
    g = golden
    st = dict(vel=g["g5_vel0"].copy(), old_time=0.0, time_zero=0, first=True)
    for m, want in zip(g["g5_msgs"], g["g5_states"]):
        st = eo.imu_step(st, int(m[0]), int(m[1]), m[2:6], m[6:9], g["g5_cov_diag"], m[9:12])
        got = np.concatenate([st["vel"], [st["old_time"], st["time_zero"]], st["rotation"].ravel(), st["normal"],
                              st["ang"], st["ang_err"]])
        close(got, want, 1e-13)
    close(golden["g5_states"][1][:3], [0.09819262, 0.1123517, 0.0593944], 1e-7)


@pytest.mark.parametrize("lvl", [0, 1, 2])
def test_g6_of_simulation_injected(golden, lvl):
    g = golden
    v_obs, feas, Rb = eo.of_simulation(g["g1_v"], g["g1_omega"], 1, g["g1_n"], g["g1_t"], g["g1_points"], g["g1_flow"],
                                       g[f"g6_{lvl}_sig"], g[f"g6_{lvl}_z"], 16)
    close(v_obs, g[f"g6_{lvl}_v_obs"], 1e-10); close(Rb, g[f"g6_{lvl}_bound"], 1e-9)
    close(feas, g[f"g6_{lvl}_feasible_last"], 1e-10)


def test_g7_pix_trans_static(golden):
    g = golden
    for i, o in zip(g["g7_in"], g["g7_out"]):
        close(eo.pix_trans(tuple(int(v) for v in i)), o)
    assert np.array_equal(eo.static_immobile(g["g7_newpos"], g["g7_oldpos"], 3.0, 1.5, -1.0), g["g7_static"])


def test_g9_of_module_system(golden):
    g = golden
    v, R, rank, s = eo.solve_of_module(g["g9_x"], g["g9_u"], g["g9_dist"], [0, 0, 1])
    close(v, g["g9_v"], 1e-10); close(R, g["g9_R"], 1e-9); close(s, g["g9_s"], 1e-11); assert rank == int(g["g9_rank"])


def test_g10_post_solve(golden):
    g = golden
    close(eo.post_solve(g["g10_v_obs"], g["g10_rotation"], g["g10_ang"], g["g10_offset"]), g["g10_v_uav"], 1e-13)


def test_kf_scalar_recursion():
    """of_module.py:63-76 matrices (F=B=H=I, Q=1e-5 I, R=10 I, P0=.1 I) reduce to a scalar recursion per axis
    (SURVEY.md Appendix B5).  Parity unpinned (OpenCV absent): closed-form KAT only."""
    I = np.eye(3)
    x = np.zeros(3); P = 0.1 * I
    xs, ps = 0.0, 0.1
    rng = np.random.default_rng(0)
    for _ in range(25):
        u = rng.normal(0, 0.01, 3); z = rng.normal(0, 1, 3)
        x, P = eo.kf_predict(x, P, I, 1e-5 * I, I, u)
        x, P = eo.kf_correct(x, P, I, 10 * I, z)
        xs = xs + u[0]; ps = ps + 1e-5
        k = ps / (ps + 10); xs = xs + k * (z[0] - xs); ps = (1 - k) * ps
        assert abs(x[0] - xs) < 1e-14 and abs(P[0, 0] - ps) < 1e-15
        assert abs(P[0, 1]) < 1e-18


# ---- feas_simulation / overlap (simulation.py:70-104, 124-136): golden vectors from the reference's own functions driven like its
#      live experiment (simulation.py:753-774), tests/golden/make_golden_feas.py
@pytest.fixture(scope="module")
def feas_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_feas.npz"))


@pytest.mark.parametrize("case", [0, 1])
def test_feas_simulation_oracle_matches_reference(feas_golden, case):
    g = feas_golden
    tr, sig, z = g[f"f{case}_truth"], g[f"f{case}_sig"], g[f"f{case}_z"]
    mean, v_obs = eo.feas_simulation(tr[3:6], tr[6], tr[7:10], tr[10:13], g[f"f{case}_pos"], g[f"f{case}_true_flow"], tr[13:16], sig, z, len(z))
    np.testing.assert_allclose(mean, g[f"f{case}_mean"], rtol=1e-10, atol=1e-12)
    a, b = g[f"f{case}_split"]
    for q in range(6):
        r = g[f"f{case}_mean"][q]
        assert [eo.overlap(r[:a], r[a:b]), eo.overlap(r[a:b], r[b:]), eo.overlap(r[:a], r[b:])] == list(g[f"f{case}_overlap"][q])


def test_overlap_oracle_matches_reference(feas_golden):
    g = feas_golden
    assert eo.overlap(g["ov_d1"], g["ov_d2"]) == int(g["ov_12"]) and eo.overlap(g["ov_d3"], g["ov_d3"]) == int(g["ov_33"]) == 40
    assert eo.overlap(g["ov_d1"], g["ov_d3"]) == int(g["ov_13"])


def test_sweep_steps_follow_the_reference_blocks(pkg):
    """simulation.sweep_step restates what step i of each of the eight sweep blocks (simulation.py:183-461) feeds of_simulation;
    and the oracle's of_simulation on those inputs lands on the curve the reference saved (CPU, two axes, 150 trials)."""
    import os
    import of_amd.simulation as sim
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_sweeps.npz"))
    raw = g["points_raw"]
    d, h, n, sig = sim.sweep_step("flow_errors", 40, 100, raw, 1.0, [0, 0, 1], None)
    np.testing.assert_allclose(d[:, 0], (raw[:, 0] - raw[:, 0].mean()) * 1.27); np.testing.assert_allclose(d[:, 1], (raw[:, 1] - raw[:, 1].mean()) * 0.93)
    assert sig[3] == 0.04 and abs(sig[4] - np.sqrt(2) * 0.04) < 1e-15 and sig[0] == 0.00071 and h == 1.0
    d, h, n, sig = sim.sweep_step("point_position", 30, 100, raw, 1.0, [0, 0, 1], None)
    np.testing.assert_allclose(d[:, 0], raw[:, 0] - raw[:, 0].mean() * 1.27 + 0.3)       # the block's own precedence (:443) and shift (:451)
    d, h, n, sig = sim.sweep_step("height", 99, 100, raw, 1.0, [0, 0, 1], None)
    assert abs(h - 7.85) < 1e-12
    d, h, n, sig = sim.sweep_step("orientation", 50, 100, raw, 1.0, [0, 0, 1], None)
    np.testing.assert_allclose(n, [1.0, 0.0, 0.0], atol=1e-15)
    for axis, i in (("distance_error", 60), ("translation_error", 60), ("ang_vel_error", 60), ("normal_error", 60)):
        sig = sim.sweep_step(axis, i, 100, raw, 1.0, [0, 0, 1], None)[3]
        assert sig[{"distance_error": 2, "translation_error": 1, "ang_vel_error": 0, "normal_error": 5}[axis]] == 0.06
    rng = np.random.default_rng(4)
    v, om, t = np.array([1.0, 1, 1]), np.array([1.0, 1, 1]), np.array([0.02, 0, 0.205])
    for axis, i in (("flow_errors", 70), ("point_position", 50)):
        saved = g[sim.SWEEP_AXES[axis][0]]
        mean_s, std_s = saved[:300].reshape(100, 3)[i], saved[300:].reshape(100, 3)[i]
        d, h, n, sig = sim.sweep_step(axis, i, 100, raw, 1.0, [0, 0, 1], None)
        tf = eo.generate_test_data(d, v, om, h, n, t)
        T = 150
        vo, _, _ = eo.of_simulation(v, om, h, n, t, d, tf, sig, rng.standard_normal((T, 10 + 4 * len(d))), T)
        assert np.all(np.abs(vo.mean(0) - mean_s) < 4 * std_s / 10 + 4 * vo.std(0) / np.sqrt(T)), (axis, vo.mean(0), mean_s)
        assert np.all(np.abs(vo.std(0) / std_s - 1) < 0.4), (axis, vo.std(0), std_s)


def test_philox_known_answers_and_noise_rows():
    """The counter-based noise of the Monte-Carlo sweeps (oracle/estimation_oracle.py noise_normals = csrc/k_estimate.hip
    ofk_noise_normal): Philox4x32-10 against the known-answer vectors of Random123 (Salmon et al., SC'11: kat_vectors, philox4x32 10),
    the normals' moments, and the independence of rows from how many trials are generated (what sharding relies on)."""
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = eo.philox4x32_10([ctr[0]], [ctr[1]], [ctr[2]], [ctr[3]], key[0], key[1])
        assert tuple(int(x[0]) for x in got) == want
    z = eo.noise_normals(99, 3, 5, 200001)
    assert z.shape == (200001,) and abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.std() - 1) < 0.01 and abs((z ** 4).mean() - 3) < 0.1
    assert np.array_equal(eo.noise_normals(99, 3, 5, 7), z[:7])                       # a prefix is a prefix (odd counts too)
    rows = eo.noise_rows(99, 3, 4, 3, 50)
    assert rows.shape == (3, 210) and np.array_equal(rows[1], eo.noise_normals(99, 3, 5, 210))
    assert not np.array_equal(eo.noise_normals(99, 4, 5, 16), z[:16]) and not np.array_equal(eo.noise_normals(98, 3, 5, 16), z[:16])
