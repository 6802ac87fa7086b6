"""oracle/estimation_oracle.py — numpy restatement of the reference's estimation math.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports it.

Parity PINNED: every function below follows the cited reference lines and is checked in
tests/test_oracle_golden.py against tests/golden/reference_numpy.npz, which was produced
by running the reference's own functions (tests/golden/make_golden.py).  The Kalman
filter (cv2.KalmanFilter, of_module.py:63-76) is the exception: OpenCV is absent, so
`kf_predict` / `kf_correct` restate the textbook recursion —
parity unpinned for that one.

The per-point Python loops of the reference are vectorised; rows are stacked in the same
order and the same `np.linalg.lstsq` (LAPACK gelsd) call is made, so results agree with
the reference to rounding (checked at 1e-12).
"""
import numpy as np


def pix_trans(img_dim):
    """of_library.py:31-43 — centre offset per axis: d/2 if even else (d+1)/2 (true division, py3)."""
    tx = img_dim[0] / 2 if img_dim[0] % 2 == 0 else (img_dim[0] + 1) / 2
    ty = img_dim[1] / 2 if img_dim[1] % 2 == 0 else (img_dim[1] + 1) / 2
    return tx, ty


def _hom(x):
    x = np.asarray(x, dtype=np.float64)
    return np.concatenate([x, np.ones((len(x), 1))], axis=1)


def generate_test_data(x, v, omega, d, n, t=None):
    """velocity_measurment_node:25-29 (t=None) / simulation.py:7-12 (lever arm t)."""
    v = np.asarray(v, dtype=np.float64)
    omega = np.asarray(omega, dtype=np.float64)
    n = np.asarray(n, dtype=np.float64)
    if t is not None:
        v = v + np.cross(omega, t)
    p = _hom(x)
    wxp = np.cross(omega[None, :], p)
    flow = (p @ n)[:, None] / d * (v[None, :] - v[2] * p) + (wxp - wxp[:, 2:3] * p)
    return flow[:, :2]


def _xhat(x):
    """Stack of [p]x matrices, (N,3,3): [[0,-1,y],[1,0,-x],[-y,x,0]] (node :35)."""
    x = np.asarray(x, dtype=np.float64)
    N = len(x)
    X = np.zeros((N, 3, 3))
    X[:, 0, 1] = -1.0; X[:, 0, 2] = x[:, 1]
    X[:, 1, 0] = 1.0;  X[:, 1, 2] = -x[:, 0]
    X[:, 2, 0] = -x[:, 1]; X[:, 2, 1] = x[:, 0]
    return X


def _system(x, u, n, omega):
    x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
    X = _xhat(x)
    u3 = np.concatenate([u[:, :2], np.zeros((len(x), 1))], axis=1)
    inner = u3 + np.einsum("nij,j->ni", X, np.asarray(omega, dtype=np.float64))
    b = np.einsum("nij,nj->ni", X, inner)
    ndotp = _hom(x) @ np.asarray(n, dtype=np.float64)
    return X, b, ndotp


def solve_lgs_node(x, u, d, n, omega):
    """velocity_measurment_node:30-42 — A_i=[p]x, b_i=[p]x(u+[p]x w)/(n.p); lstsq(A, B*d) -> (v,R,rank,s)."""
    X, b, ndotp = _system(x, u, n, omega)
    A = X.reshape(-1, 3)
    B = (b / ndotp[:, None]).reshape(-1)
    return np.linalg.lstsq(A, B * d, rcond=None)


def solve_lgs_sim(x, u, d, n, omega, t):
    """simulation.py:15-30 — A_i=[p]x (n.p), b_i=[p]x(u+[p]x w); returns (v - w x t, R, s)."""
    X, b, ndotp = _system(x, u, n, omega)
    A = (X * ndotp[:, None, None]).reshape(-1, 3)
    v, R, rank, s = np.linalg.lstsq(A, b.reshape(-1) * d, rcond=None)
    return v - np.cross(omega, t), R, s


def solve_lgs_eval(x, u, d, n, omega, t):
    """evaluate_exp.py:18-31 — node system, then v - w x t; returns (v, R)."""
    v, R, rank, s = solve_lgs_node(x, u, d, n, omega)
    return v - np.cross(omega, t), R


def solve_of_module(x3, u3, dist, n):
    """of_module.py:139-146 — A_i=[p]x/dist_i, b_i=A_i u_i/(n.p) (u is a 3-vector, no omega term)."""
    x3 = np.asarray(x3, dtype=np.float64); u3 = np.asarray(u3, dtype=np.float64)
    Ai = _xhat(x3[:, :2]) / np.asarray(dist, dtype=np.float64)[:, None, None]
    bi = np.einsum("nij,nj->ni", Ai, u3) / (x3 @ np.asarray(n, dtype=np.float64))[:, None]
    return np.linalg.lstsq(Ai.reshape(-1, 3), bi.reshape(-1), rcond=None)


def r_tilde(x, u, n, v, dist):
    """of_library.py:365-386 — feasibility cosine r and distance ratio d per point."""
    p = _hom(x)
    u3 = np.concatenate([np.asarray(u, dtype=np.float64)[:, :2], np.zeros((len(p), 1))], axis=1)
    vc = -np.cross(p, np.asarray(v, dtype=np.float64)[None, :])
    uc = np.cross(p, u3)
    vn = np.linalg.norm(vc, axis=1); un = np.linalg.norm(uc, axis=1)
    zero = (un * vn) == 0
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.einsum("ni,ni->n", vc, uc) * (1.0 / un) / vn
        pn = p @ np.asarray(n, dtype=np.float64)
        r = np.where(pn < 0, -r, r)
        dd = pn * vn * (1.0 / un) / dist
    r = np.where(zero, 1.0, r)
    dd = np.where(zero, 1.0, dd)
    return r, dd


def r_tilde_legacy(x3, u3, n, v):
    """sensor_precision_experiments/pixhawk_pure_IMU/of_library.py:365-380 — 4-arg form on homogeneous
    3-vectors, no zero guard, no /dist (the variant of_module.py:125 calls)."""
    x3 = np.asarray(x3, dtype=np.float64); u3 = np.asarray(u3, dtype=np.float64)
    vc = -np.cross(x3, np.asarray(v, dtype=np.float64)[None, :])
    uc = np.cross(x3, u3)
    vn = np.linalg.norm(vc, axis=1); iun = 1.0 / np.linalg.norm(uc, axis=1)
    r = np.einsum("ni,ni->n", vc, uc) * iun / vn
    pn = x3 @ np.asarray(n, dtype=np.float64)
    r = np.where(pn < 0, -r, r)
    return r, pn * vn * iun


def feasibility_sim(position, linear_velocity, flow, angular_velocity, translation, normal):
    """simulation.py:108-120 — returns array([parallelity, length])."""
    p = _hom(position)
    w = np.asarray(angular_velocity, dtype=np.float64)
    f3 = np.concatenate([np.asarray(flow, dtype=np.float64)[:, :2], np.zeros((len(p), 1))], axis=1)
    fac1 = np.cross(p, (np.asarray(linear_velocity, dtype=np.float64) - np.cross(w, translation))[None, :])
    fac2 = np.cross(p, f3 - np.cross(w[None, :], p))
    n1 = np.linalg.norm(fac1, axis=1); n2 = np.linalg.norm(fac2, axis=1)
    par = np.einsum("ni,ni->n", fac1, fac2) / (n1 * n2)
    length = n1 / n2 * (p @ np.asarray(normal, dtype=np.float64))
    return np.array([par, length])


def static_immobile(newpos, oldpos, maxspeed, distance, dummy_value):
    """of_library.py:88-92."""
    speed = np.abs(newpos - oldpos) < (maxspeed / distance)
    dummy = oldpos != dummy_value
    stable = speed * dummy
    return stable[:, :, 0] * stable[:, :, 1]


def quat_to_rot(qx, qy, qz, qw):
    """velocity_measurment_node:66-68."""
    return np.array([[1.0 - 2 * (qy ** 2 + qz ** 2), 2 * (qx * qy - qw * qz), 2 * (qw * qy + qx * qz)],
                     [2 * (qx * qy + qw * qz), 1.0 - 2 * (qx ** 2 + qz ** 2), 2 * (qy * qz - qw * qx)],
                     [2 * (qx * qz - qw * qy), 2 * (qw * qx + qy * qz), 1.0 - 2 * (qx ** 2 + qy ** 2)]])


def imu_step(state, secs, nsecs, q, ang, ang_cov_diag, acc):
    """velocity_measurment_node:61-89 (call_imu).  `state` is a dict with vel, old_time, time_zero, first;
    returns the updated dict (+ rotation, normal, ang, ang_err).  got_vel_ is always False (node :260-262
    toggles a local)."""
    st = dict(state)
    st["ang"] = np.asarray(ang, dtype=np.float64)
    st["ang_err"] = np.asarray(ang_cov_diag, dtype=np.float64)
    R = quat_to_rot(*q)
    st["rotation"] = R
    st["normal"] = R @ np.array([0.0, 0.0, 1.0])
    if st["first"]:
        st["old_time"] = float(nsecs) / 10 ** 9
        st["time_zero"] = secs
        st["first"] = False
    else:
        current = float(secs - st["time_zero"]) + float(nsecs) / 10 ** 9
        elapsed = current - st["old_time"]
        st["vel"] = st["vel"] + R @ (np.asarray(acc, dtype=np.float64) - 9.81 * st["normal"]) * elapsed
        st["old_time"] = float(current)
    return st


def post_solve(v_obs, rotation, ang, offset):
    """velocity_measurment_node:258 — v_uav = R (v_obs - [w]x offset)."""
    W = np.array([[0, -ang[2], ang[1]], [ang[2], 0, -ang[0]], [-ang[1], ang[0], 0]], dtype=np.float64)
    return rotation @ (np.asarray(v_obs, dtype=np.float64) - W @ np.asarray(offset, dtype=np.float64))


# ---- Kalman filter (cv2.KalmanFilter semantics, float64 as the reference assigns float64 matrices;
#      of_module.py:63-76,122,152).  PARITY UNPINNED (OpenCV absent).
def kf_predict(x, P, F, Q, B=None, u=None):
    x = F @ x
    if B is not None and u is not None:
        x = x + B @ u
    P = F @ P @ F.T + Q
    return x, P


def kf_correct(x, P, H, Rm, z):
    S = H @ P @ H.T + Rm
    K = np.linalg.solve(S, H @ P).T
    x = x + K @ (z - H @ x)
    P = P - K @ H @ P
    return x, P


def of_simulation(linear_velocity, angular_velocity, height_above_gr, normal_vector, translation, pos, true_flow,
                  sig, z, iterations):
    """simulation.py:36-66 with the np.random.normal draws replaced by a flat tensor of standard normals `z`
    consumed in the reference's draw order: per iteration 3 (omega), 3 (t), 1 (height), 2N (flow),
    2N (position), 3 (normal; drawn then discarded, simulation.py:45-46).
    sig = (ang_vel_sig, translation_sig, height_sig, flow_sig, position_sig, normal_sig).
    Returns v_obs (iterations,3), analytic bound R (iterations,), last feasibility (2,N)."""
    N = len(pos)
    per = 3 + 3 + 1 + 2 * N + 2 * N + 3
    z = np.asarray(z, dtype=np.float64).reshape(iterations, per)
    v_obs = np.zeros((iterations, 3)); Rb = np.zeros(iterations)
    lv = np.asarray(linear_velocity, dtype=np.float64); av = np.asarray(angular_velocity, dtype=np.float64)
    nv = np.asarray(normal_vector, dtype=np.float64); tr = np.asarray(translation, dtype=np.float64)
    feas = None
    for i in range(iterations):
        zi = z[i]; o = 0
        ang_err = av + sig[0] * zi[o:o + 3]; o += 3
        tr_err = tr + sig[1] * zi[o:o + 3]; o += 3
        h_err = height_above_gr + sig[2] * zi[o:o + 1]; o += 1
        flow_err = true_flow + sig[3] * zi[o:o + 2 * N].reshape(N, 2); o += 2 * N
        pos_err = pos + sig[4] * zi[o:o + 2 * N].reshape(N, 2); o += 2 * N
        normal_err = nv / np.linalg.norm(nv)
        v_obs[i], Res, singular = solve_lgs_sim(pos_err, flow_err, h_err, normal_err, ang_err, tr_err)
        xp = _hom(pos)
        dxp = _hom(pos_err) - xp
        ddotx = np.concatenate([flow_err - true_flow, np.zeros((N, 1))], axis=1)
        v_err = (h_err - height_above_gr) / height_above_gr * (xp @ nv) + xp @ (normal_err - nv) + dxp @ nv
        d_err = ddotx + np.cross(dxp, av[None, :]) + np.cross(xp, (ang_err - av)[None, :]) + dxp
        part = np.linalg.norm(np.cross(xp, v_err[:, None] * lv[None, :] + height_above_gr * d_err), axis=1) / np.amin(singular)
        Rb[i] = np.sqrt(np.sum(part ** 2)) + np.linalg.norm(av) * sig[1] + sig[0] * np.linalg.norm(tr) + sig[0] * sig[1]
        feas = feasibility_sim(pos_err, lv, flow_err, ang_err, tr_err, normal_err)
    return v_obs, feas, Rb


def feas_simulation(angular_velocity, height_above_gr, normal_vector, translation, pos, true_flow, true_vel, sig, z, iterations,
                    per_trial=False):
    """simulation.py:70-104 with the np.random draws replaced by standard normals `z` in the reference's draw order: per
    iteration 3 (omega), 3 (t), 1 (height), 2N (flow), 2N (position), 3 (velocity), 1 + 1 (the two orientation errors, each
    normal_sig * N(0, normal_sig), :87-88).  sig = (ang_vel, translation, height, flow, position, normal, velocity) — the
    reference reads the last two from module globals.  Returns the six per-point means in the reference's order
    (backward_para, backward_dist, forward_para, forward_dist, backward_res, forward_res) as [6, N] and v_obs [iterations, 3]
    (with per_trial=True also the [iterations, 6, N] table)."""
    N = len(pos)
    per = 12 + 4 * N
    z = np.asarray(z, dtype=np.float64).reshape(iterations, per)
    av = np.asarray(angular_velocity, np.float64); nv = np.asarray(normal_vector, np.float64)
    tr = np.asarray(translation, np.float64); tv = np.asarray(true_vel, np.float64)
    table = np.zeros((iterations, 6, N)); v_obs = np.zeros((iterations, 3))
    for i in range(iterations):
        zi = z[i]; o = 0
        ang_err = av + sig[0] * zi[o:o + 3]; o += 3
        tr_err = tr + sig[1] * zi[o:o + 3]; o += 3
        h_err = height_above_gr + sig[2] * zi[o:o + 1]; o += 1
        flow_err = true_flow + sig[3] * zi[o:o + 2 * N].reshape(N, 2); o += 2 * N
        pos_err = pos + sig[4] * zi[o:o + 2 * N].reshape(N, 2); o += 2 * N
        vel_err = tv + sig[6] * zi[o:o + 3]; o += 3
        o1 = sig[5] * (sig[5] * zi[o]); o2 = sig[5] * (sig[5] * zi[o + 1])
        Ry = np.array([[np.cos(o2), 0, np.sin(o2)], [0, 1, 0], [-np.sin(o2), 0, np.cos(o2)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(o1), -np.sin(o1)], [0, np.sin(o1), np.cos(o1)]])
        normal_err = Ry @ Rx @ nv
        v_obs[i], _, _ = solve_lgs_sim(pos_err, flow_err, h_err, normal_err, ang_err, tr_err)
        table[i, 0], table[i, 1] = feasibility_sim(pos_err, v_obs[i], flow_err, ang_err, tr_err, normal_err)
        table[i, 2], table[i, 3] = feasibility_sim(pos_err, vel_err, flow_err, ang_err, tr_err, normal_err)
        # rows of the system (:96-101): A_j = [p]x (n.p), b_j = [p]x (u + [p]x omega)
        xh = _xhat(pos_err)                                                     # [N,3,3]
        u3 = np.concatenate([flow_err, np.zeros((N, 1))], axis=1)
        b = np.einsum("nij,nj->ni", xh, u3 + np.einsum("nij,j->ni", xh, ang_err))
        A = xh * (_hom(pos_err) @ normal_err)[:, None, None]
        table[i, 4] = np.linalg.norm(np.einsum("nij,j->ni", A, v_obs[i]) - b, axis=1)
        table[i, 5] = np.linalg.norm(np.einsum("nij,j->ni", A, vel_err) - b, axis=1)
    mean = np.mean(table, axis=0)
    return (mean, v_obs, table) if per_trial else (mean, v_obs)


def overlap(data1, data2, bins=100):
    """simulation.py:124-136 — histogram both samples over the common edges of the stacked sample, sum of the bin-wise minima."""
    edges = np.histogram(np.hstack((data1, data2)), bins=bins)[1]
    return int(np.sum(np.minimum(np.histogram(data1, bins=edges)[0], np.histogram(data2, bins=edges)[0])))


def associate(t_img, imu_t, imu_q, imu_w, hgt_t, hgt_r):
    """evaluate_exp.py:77-95 for a batch of image times: nearest IMU / range sample (np.argmin of the absolute time
    difference: the first minimum), dist = range, R from the quaternion (same expression as quat_to_rot), normal = R e_z,
    omega = angular velocity.  Returns (imu_index, hgt_index, d, R [n,3,3], normal [n,3], omega [n,3])."""
    t_img = np.atleast_1d(np.asarray(t_img, np.float64))
    imu_t = np.asarray(imu_t, np.float64); hgt_t = np.asarray(hgt_t, np.float64)
    imu_q = np.asarray(imu_q, np.float64).reshape(-1, 4); imu_w = np.asarray(imu_w, np.float64).reshape(-1, 3)
    ii = np.array([int(np.argmin(np.abs(imu_t - t))) for t in t_img])
    hi = np.array([int(np.argmin(np.abs(hgt_t - t))) for t in t_img])
    R = np.array([quat_to_rot(*imu_q[k]) for k in ii])
    normal = np.array([np.dot(Rk, np.array([0, 0, 1])) for Rk in R])
    return ii, hi, np.asarray(hgt_r, np.float64)[hi], R, normal, imu_w[ii]


def d_split(d, d_exp_err):
    """velocity_measurment_node:250-252 — sorted plane distances, consecutive differences, and the split count the commented
    line :252 describes (gaps >= d_exp_err)."""
    d_sorted = np.sort(np.asarray(d, np.float64))
    d_diff = d_sorted[1:] - d_sorted[:-1]
    return d_sorted, d_diff, int(np.sum(d_diff >= d_exp_err))


def feature_eval(pos, pos_err, oldpos, oldpos_err, vel, vel_err, focal_len, dummy_value, img_dim, weight):
    """of_library.py:270-286 (calc_height), :53-75 + :100-114 (convert_to_of inside dynamic_immobile), :291-317 (eval_ft) for ONE
    track set, chained the way initialize_ft (:231-263) uses them.  PARITY UNPINNED: the reference's three functions carry
    undefined names and cannot run; this restates the formulas they spell out.  pos, oldpos [n,2] px; pos_err,
    oldpos_err [n]; vel, vel_err [3].  Returns (height, height_err, immobile, score, order)."""
    pos = np.asarray(pos, np.float64).reshape(-1, 2); oldpos = np.asarray(oldpos, np.float64).reshape(-1, 2)
    e = np.asarray(pos_err, np.float64).reshape(-1); oe = np.asarray(oldpos_err, np.float64).reshape(-1)
    vx, vy, vz = [float(v) for v in vel]; vex, vey, vez = [float(v) for v in vel_err]
    f = float(focal_len)
    tx, ty = pix_trans(img_dim)
    px, py = pos[:, 0], pos[:, 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        ofx, ofy = px - oldpos[:, 0], py - oldpos[:, 1]
        nx, ny = f * vx - px * vz, f * vy - py * vz
        hx, hy = nx / ofx, ny / ofy
        hxe = (f * vex / ofx) ** 2 + (nx * e / (ofx * ofx)) ** 2 + (e * vz / ofx) ** 2 + (px * vez / ofx) ** 2
        hye = (f * vey / ofy) ** 2 + (ny * e / (ofy * ofy)) ** 2 + (e * vz / ofy) ** 2 + (py * vez / ofy) ** 2
        h = 0.5 * (hx + hy); he = hxe + hye
        xe = (f - (px - tx) / h) * vx / h; ye = (f - (py - ty) / h) * vy / h
        fx, fy = f - px + tx, f - py + ty
        xee = (e * vx / h) ** 2 + (fx * vex / h) ** 2 + (fx * vx * he / (h * h)) ** 2
        yee = (e * vy / h) ** 2 + (fy * vey / h) ** 2 + (fy * vy * he / (h * h)) ** 2
        obs_err = oe * oe + e * e
        immobile = ((ofx - xe) ** 2 < obs_err + xee) & ((ofy - ye) ** 2 < obs_err + yee) & (oldpos[:, 0] != dummy_value) & \
            (oldpos[:, 1] != dummy_value)

        def norm(a):
            rng_ = np.amax(a) - np.amin(a)
            return (a - np.amin(a)) / rng_ if rng_ > 0 else np.zeros_like(a)
        quad = (px - tx) ** 2 + (py - ty) ** 2
        dn = quad / np.amax(quad) if np.amax(quad) > 0 else np.zeros_like(quad)
        score = weight[0] * (1 - norm(h)) + weight[1] * norm(he) + weight[2] * (1 - dn) + weight[3] * norm(e)
    return h, he, immobile, score, np.argsort(score, kind="stable")


# ------------------------------------------------------------------------------------------------ counter-based noise (Monte-Carlo sweeps)
# The reference draws its noise with np.random.normal, unseeded (simulation.py:40-47); the golden vectors inject those draws.  For the
# 4096-wide batches of BASELINE configs[4] the product can draw the normals on the device instead (csrc/k_estimate.hip ofk_noise_normal,
# include/ofk.h ofk_of_simulation_rng); this is the same function in numpy.  The integer part is Philox4x32-10 (Salmon et al., "Parallel
# random numbers: as easy as 1, 2, 3", SC'11 - Random123), pinned by that paper's known-answer vectors (tests/test_oracle_golden.py).
_PH_M0, _PH_M1, _PH_W0, _PH_W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on arrays of 32-bit counter words (uint64 holders) -> four arrays of 32-bit outputs."""
    c = [np.asarray(x, np.uint64) & _U32 for x in (c0, c1, c2, c3)]
    k = [np.uint64(int(k0) & 0xFFFFFFFF), np.uint64(int(k1) & 0xFFFFFFFF)]
    for _ in range(10):
        p0 = _PH_M0 * c[0]; p1 = _PH_M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & _U32, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & _U32]
        k = [(k[0] + _PH_W0) & _U32, (k[1] + _PH_W1) & _U32]
    return c


def noise_normals(seed, step, trial, count):
    """Elements 0 .. count - 1 of the noise row of (seed, step, trial): element e = output (e & 1) of the Box-Muller transform of
    Philox4x32-10(counter (e >> 1, trial, step, 0), key (seed low, seed high))."""
    j = np.arange((int(count) + 1) // 2, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32_10(j, np.full_like(j, int(trial)), np.full_like(j, int(step)), np.zeros_like(j), int(seed) & 0xFFFFFFFF, int(seed) >> 32)
    u1 = ((x0 >> np.uint64(5)).astype(np.float64) * 67108864.0 + (x1 >> np.uint64(6)).astype(np.float64) + 0.5) * 2.0 ** -53
    u2 = ((x2 >> np.uint64(5)).astype(np.float64) * 67108864.0 + (x3 >> np.uint64(6)).astype(np.float64) + 0.5) * 2.0 ** -53
    r = np.sqrt(-2.0 * np.log(u1)); th = 6.283185307179586476925 * u2
    out = np.empty(2 * len(j), np.float64)
    out[0::2] = r * np.cos(th); out[1::2] = r * np.sin(th)
    return out[:int(count)]


def noise_rows(seed, step, trial0, trials, n_points):
    """[trials, 10 + 4 n] normals: what ofk_of_simulation_rng draws for trials trial0 .. trial0 + trials - 1 (of_simulation's z)."""
    return np.stack([noise_normals(seed, step, trial0 + t, 10 + 4 * int(n_points)) for t in range(int(trials))])
