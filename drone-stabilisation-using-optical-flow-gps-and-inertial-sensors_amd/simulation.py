"""simulation.py — drop-in for the functions of numerical_simulation/simulation.py, batched on the GPU.

Same names and positional signatures: generate_test_data (:7-12), solve_lgs (:15-30), feasibility (:108-120),
of_simulation (:36-66), feas_simulation (:70-104), overlap (:124-136).  The reference's module-level globals `iterations`,
`true_flow`, `normal_sig` and `velocity_sig` (read by of_simulation / feas_simulation) are module attributes here as well.  Noise: the reference draws from an unseeded np.random; here
the standard-normal tensor is drawn on the host in the reference's draw order (optionally seeded, or passed in)
and all `iterations` trials are solved by one launch of k_of_simulation.
"""
import numpy as np

try:
    from . import ofk
except ImportError:
    import ofk

iterations = 100
true_flow = None
normal_sig = 0.00065      # simulation.py:171 (feas_simulation reads it as a global; its parameter is misspelt `normal_sigi`, :70)
velocity_sig = 0.01       # simulation.py:172
rng = None            # set to np.random.default_rng(seed) for reproducible sweeps


def generate_test_data(x, v, omega, d, n, t):
    return ofk.default_context().flow_model(np.asarray(x, np.float64)[:, :2], v, omega, float(np.ravel(d)[0]), n, t)


def solve_lgs(x, u, d, n, omega, t):
    """Returns (v - omega x t, R, s) like the reference; R has shape (1,) (empty when rank < 3 or N < 2)."""
    try:
        x = np.asarray(x, np.float64)
        out = ofk.default_context().velocity_solve(ofk.SOLVE_SIM, x[:, :2], np.asarray(u, np.float64)[:, :2],
                                                   d=float(np.ravel(d)[0]), nrm=n, omega=omega, t=t)
        rank = int(out[4])
        R = np.array([out[3]]) if (rank == 3 and 3 * len(x) > 3) else np.empty(0)
        return out[:3].copy(), R, out[5:8].copy()
    except ofk.OfkError:
        raise
    except Exception:     # reference: bare except around lstsq (simulation.py:29-30)
        return np.zeros(3), 10000 * np.ones(3 * len(x)), np.array([0, 0, 0])


def feasibility(position, linear_velocity, flow, angular_velocity, translation, normal):
    r, length = ofk.default_context().feasibility(ofk.FEAS_SIM, np.asarray(position, np.float64)[:, :2],
                                                  np.asarray(flow, np.float64)[:, :2], normal, linear_velocity,
                                                  omega=angular_velocity, t=translation)
    return np.array([r, length])


def draw_noise(n_points, trials, generator=None):
    g = generator or rng or np.random.default_rng()
    return g.standard_normal((trials, 10 + 4 * n_points))


def of_simulation(linear_velocity, angular_velocity, height_above_gr, normal_vector, translation, pos, ang_vel_sig,
                  translation_sig, height_sig, flow_sig, position_sig, normal_sig, z=None):
    """Returns (v_obs [iterations,3], feasible [2,N] of the last trial, R [iterations]) like the reference."""
    pos = np.asarray(pos, np.float64)
    tf = np.asarray(true_flow, np.float64)
    if z is None:
        z = draw_noise(len(pos), iterations)
    z = np.asarray(z, np.float64).reshape(-1, 10 + 4 * len(pos))
    truth = np.concatenate([np.asarray(linear_velocity, np.float64), np.asarray(angular_velocity, np.float64),
                            [float(height_above_gr)], np.asarray(normal_vector, np.float64), np.asarray(translation, np.float64)])
    sig = np.array([ang_vel_sig, translation_sig, height_sig, flow_sig, position_sig, normal_sig], np.float64)
    v_obs, bound = ofk.default_context().of_simulation(truth, sig, pos, tf, z)
    # feasibility of the last trial's perturbed inputs (simulation.py:65)
    zl = z[-1]; N = len(pos)
    ang = np.asarray(angular_velocity, np.float64) + sig[0] * zl[0:3]
    tr = np.asarray(translation, np.float64) + sig[1] * zl[3:6]
    fl = tf + sig[3] * zl[7:7 + 2 * N].reshape(N, 2)
    pe = pos + sig[4] * zl[7 + 2 * N:7 + 4 * N].reshape(N, 2)
    nv = np.asarray(normal_vector, np.float64)
    feas = feasibility(pe, linear_velocity, fl, ang, tr, nv / np.linalg.norm(nv))
    return v_obs, feas, bound


def draw_feas_noise(n_points, trials, generator=None):
    """Standard normals for feas_simulation in the reference's draw order: 3 + 3 + 1 + 2N + 2N + 3 + 1 + 1 per trial."""
    g = generator or rng or np.random.default_rng()
    return g.standard_normal((trials, 12 + 4 * n_points))


def feas_simulation(linear_velocity, angular_velocity, height_above_gr, normal_vector, translation, pos, ang_vel_sig, translation_sig,
                    height_sig, flow_sig, position_sig, normal_sigi, true_vel, z=None):
    """simulation.py:70-104: per trial perturbed inputs -> solve -> backward / forward feasibility -> per-point residual norms;
    returns the six per-point means (backward_para, backward_dist, forward_para, forward_dist, backward_res, forward_res).
    All `iterations` trials run in one launch of k_feas_simulation.  Like the reference, the function reads `true_flow`,
    `iterations`, `normal_sig` and `velocity_sig` from the module (its `normal_sigi` parameter is unused there too) and does not
    use `linear_velocity`; unlike the reference it takes any number of points (the reference hard-codes 200, :83-84)."""
    pos = np.asarray(pos, np.float64)
    tf = np.asarray(true_flow, np.float64)
    if z is None:
        z = draw_feas_noise(len(pos), iterations)
    z = np.asarray(z, np.float64).reshape(-1, 12 + 4 * len(pos))
    truth = np.concatenate([np.asarray(linear_velocity, np.float64), np.asarray(angular_velocity, np.float64), [float(np.ravel(height_above_gr)[0])],
                            np.asarray(normal_vector, np.float64), np.asarray(translation, np.float64), np.asarray(true_vel, np.float64)])
    sig = np.array([ang_vel_sig, translation_sig, height_sig, flow_sig, position_sig, normal_sig, velocity_sig], np.float64)
    mean, _ = ofk.default_context().feas_simulation(truth, sig, pos, tf, z)
    return tuple(mean[k].copy() for k in range(6))


def overlap(data1, data2):
    """simulation.py:124-136: 100 common bins over both samples, sum of the bin-wise minimum counts."""
    return ofk.default_context().hist_overlap(data1, data2, 100)


# The eight Monte-Carlo sweeps of simulation.py:183-461 (each a '''-quoted block the author enabled one at a time; the saved
# effect_*.npy files are their outputs).  axis -> (file the reference saved, reference's `iterations`, what step i of k changes)
SWEEP_AXES = {
    "flow_errors":       ("effect_of_flow_errors", 100),        # :183-202  flow_sig = 0.001 i, position_sig = sqrt(2)/1000 i
    "distance_error":    ("effect_of_distance_error", 100),     # :216-235  height_sig = 0.001 i
    "ang_vel_error":     ("effect_o_ang_vel_error", 100),       # :249-271  ang_vel_sig = 0.001 i
    "normal_error":      ("effect_of_normal_error", 10),        # :285-304  normal_sig = 0.001 i (of_simulation discards the normal draw, :45-46)
    "translation_error": ("effect_of_translation_error", 100),  # :319-338  translation_sig = 0.001 i
    "orientation":       ("effect_of_orientation", 10),         # :355-377  normal = (sin(pi i/k), 0, cos(pi i/k)), true_flow regenerated
    "height":            ("effect_of_height", 100),             # :398-423  height = 0.2 + linspace(0.2, 7.65, k)[i], true_flow regenerated
    "point_position":    ("effect_of_point_position", 100),     # :441-461  points shifted by i/100 in x and y
}
SIGMA_ORDER = ("ang_vel_sig", "translation_sig", "height_sig", "flow_sig", "position_sig", "normal_sig")
DEFAULT_SIGMAS = dict(ang_vel_sig=0.00071, translation_sig=0.005, height_sig=0.01, flow_sig=0.056 * np.sqrt(2) * 1.23,
                      position_sig=0.056 * 1.23, normal_sig=0.00065)      # simulation.py:166-171


def sweep_step(axis, i, k, data, height_above_gr, normal_vector, sigmas):
    """What step i of `k` of the reference's sweep `axis` runs of_simulation with: (points, height, normal, six sigmas).
    `data` are the raw points (points.txt); every block first centres and stretches them in place (:188-189 ...), the
    point-position block with its own operator precedence (:443-444: x - mean(x) * 1.27)."""
    if axis not in SWEEP_AXES:
        raise ValueError(f"unknown sweep axis {axis!r}: one of {sorted(SWEEP_AXES)}")
    d = np.array(data, np.float64)[:, :2].copy()
    if axis == "point_position":
        d[:, 0] = d[:, 0] - np.mean(d[:, 0]) * 1.27; d[:, 1] = d[:, 1] - np.mean(d[:, 1]) * 0.93
        d = d + np.ones_like(d) * i / 100
    else:
        d[:, 0] = (d[:, 0] - np.mean(d[:, 0])) * 1.27; d[:, 1] = (d[:, 1] - np.mean(d[:, 1])) * 0.93
    sg = dict(DEFAULT_SIGMAS); sg.update(sigmas or {})
    h = float(height_above_gr); n = np.asarray(normal_vector, np.float64)
    if axis == "flow_errors":
        sg["flow_sig"] = 0.001 * i; sg["position_sig"] = np.sqrt(2) / 1000 * i
    elif axis == "distance_error":
        sg["height_sig"] = 0.001 * i
    elif axis == "ang_vel_error":
        sg["ang_vel_sig"] = 0.001 * i
    elif axis == "normal_error":
        sg["normal_sig"] = 0.001 * i
    elif axis == "translation_error":
        sg["translation_sig"] = 0.001 * i
    elif axis == "orientation":
        n = np.array([np.sin(np.pi * i / k), 0.0, np.cos(np.pi * i / k)])
    elif axis == "height":
        h = 0.2 + np.linspace(0.2, 7.65, k)[i]
    return d, h, n, [sg[name] for name in SIGMA_ORDER]


def sweep(axis, data, linear_velocity, angular_velocity, height_above_gr, normal_vector, translation, sigmas=None, k=100, trials=None,
          steps=None, generator=None, comm=None, device_seed=None):
    """One of the reference's eight Monte-Carlo sweeps (SWEEP_AXES; simulation.py:183-461): k steps, `trials` of_simulation
    iterations per step (default: the block's own `iterations`), all trials of a step in ONE launch of k_of_simulation.  Returns
    np.append(v_mean, v_std) laid out like the saved effect_*.npy files ([k,3] means then [k,3] standard deviations); with
    `steps` (a list of step indices) only those rows are computed and returned ([len(steps),3] + [len(steps),3]).

    With `comm` (sharding.Comm) the trials of every step are sharded over the ranks: rank r solves the contiguous slice
    shard_range(trials, r, world) of the step's trials on its own GPU and the per-step statistics come from one all-reduce of
    (sum v, sum v^2, count) = 7 doubles - trials never travel.  Every rank draws the step's full noise
    tensor from the same generator and keeps its slice, so the result does not depend on the number of ranks.

    device_seed (an int): the normals are drawn ON THE DEVICE by the counter-based generator of ofk_of_simulation_rng (Philox4x32-10 keyed
    by (device_seed, step index, global trial index)) instead of numpy's - no noise tensor is built or uploaded (2000 points x 4096
    trials = 262 MB per step at BASELINE configs[4]'s size), and a rank's shard is exactly the rows a single rank would draw."""
    global true_flow, iterations
    try:
        from . import sharding
    except ImportError:
        import sharding
    trials = int(trials or SWEEP_AXES[axis][1])
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    lo, hi = sharding.shard_range(trials, rank, world)
    idx = list(range(k)) if steps is None else [int(i) for i in steps]
    v_mean = np.zeros((len(idx), 3)); v_std = np.zeros((len(idx), 3))
    saved_iterations = iterations
    for row, i in enumerate(idx):
        d, h, n, sig = sweep_step(axis, i, k, data, height_above_gr, normal_vector, sigmas)
        true_flow = generate_test_data(d, linear_velocity, angular_velocity, h, n, translation)
        if device_seed is not None:
            if hi > lo:
                truth = np.concatenate([np.asarray(linear_velocity, np.float64), np.asarray(angular_velocity, np.float64), [float(h)],
                                        np.asarray(n, np.float64), np.asarray(translation, np.float64)])
                v_obs, _ = ofk.default_context().of_simulation_rng(truth, np.asarray(sig, np.float64), d, true_flow, device_seed, i, hi - lo, trial0=lo)
            else:
                v_obs = np.zeros((0, 3))
        elif hi > lo:
            z = draw_noise(len(d), trials, generator)[lo:hi]
            iterations = hi - lo
            v_obs, _, _ = of_simulation(linear_velocity, angular_velocity, h, n, translation, d, *sig, z=z)
        else:
            draw_noise(len(d), trials, generator)                # keep the generator in step with the other ranks
            v_obs = np.zeros((0, 3))
        if comm is None:
            v_mean[row] = v_obs.mean(axis=0); v_std[row] = v_obs.std(axis=0)
        else:
            v_mean[row], v_std[row], _ = sharding.combine_moments(v_obs.sum(0), (v_obs * v_obs).sum(0), len(v_obs), comm.allreduce)
    iterations = saved_iterations
    return np.append(v_mean, v_std)


def sweep_flow_errors(data, linear_velocity, angular_velocity, height_above_gr, normal_vector, translation, sigmas,
                      k=100, trials=100, generator=None, comm=None):
    """The "Effect of flow errors" driver (simulation.py:183-202) on points that are ALREADY centred / scaled (kept from round 2 for
    its callers; sweep("flow_errors", raw_points, ...) applies the block's own centring): k sigma-steps, flow_sig = 0.001 i,
    position_sig = sqrt(2)/1000 i; returns np.append(v_mean, v_std) laid out like the saved .npy files.  Sharding as in sweep()."""
    global true_flow, iterations
    try:
        from . import sharding
    except ImportError:
        import sharding
    true_flow = generate_test_data(data, linear_velocity, angular_velocity, height_above_gr, normal_vector, translation)
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    lo, hi = sharding.shard_range(trials, rank, world)
    v_mean = np.zeros((k, 3)); v_std = np.zeros((k, 3))
    for i in range(k):
        z = draw_noise(len(data), trials, generator)[lo:hi]
        if hi > lo:
            iterations = hi - lo
            v_obs, _, _ = of_simulation(linear_velocity, angular_velocity, height_above_gr, normal_vector, translation, data,
                                        sigmas["ang_vel_sig"], sigmas["translation_sig"], sigmas["height_sig"], 0.001 * i,
                                        np.sqrt(2) / 1000 * i, sigmas["normal_sig"], z=z)
        else:
            v_obs = np.zeros((0, 3))
        if comm is None:
            v_mean[i] = v_obs.mean(axis=0); v_std[i] = v_obs.std(axis=0)
        else:
            v_mean[i], v_std[i], _ = sharding.combine_moments(v_obs.sum(0), (v_obs * v_obs).sum(0), len(v_obs), comm.allreduce)
    iterations = trials
    return np.append(v_mean, v_std)
