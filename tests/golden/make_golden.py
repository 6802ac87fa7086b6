#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own numpy code.

Runs only in the build container (needs /root/reference; the GPU box has neither the
reference nor any use for this script).  Nothing of the reference is copied: its
functions are loaded by parsing the reference files with `ast`, keeping the named
FunctionDef/ClassDef nodes, and exec-ing them with numpy in scope (the files themselves
are scripts with import-time side effects: rospy imports, np.loadtxt, plt.show, an endless
spin loop).  Only inputs and the outputs those functions returned are written.

    python tests/golden/make_golden.py            # rewrites tests/golden/reference_numpy.npz

Cases (ids follow SURVEY.md Appendix C):
  G1  simulation.generate_test_data -> solve_lgs on points.txt           (simulation.py:7-30)
  G2  node generate_test_data -> solve_lgs on the 4 test features         (node :25-42, :123, :229-236)
  G3  of_library.r_tilde on G2                                            (of_library.py:365-386)
  G3b simulation.feasibility on points.txt                                (simulation.py:108-120)
  G3c legacy 4-arg r_tilde (pixhawk_pure_IMU/of_library.py:365-380)
  G4  seeded random batches through the three solve_lgs variants          (node, simulation, evaluate_exp)
  G5  optical_fusion.call_imu, scripted message sequence                  (node :61-89)
  G6  of_simulation with injected noise                                   (simulation.py:36-66)
  G7  pix_trans, static_immobile                                          (of_library.py:31-43, 88-92)
  G8  saved Monte-Carlo sweep effect_of_flow_errors.npy (statistical pin) (simulation.py:183-202)
  G9  of_module.py inline system build + lstsq                            (of_module.py:139-146)
  G10 node main-loop post-solve step                                      (node :257-258)
"""
import ast
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_numpy.npz")


def load_defs(path, names, extra_globals=None):
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    missing = set(names) - {n.name for n in keep}
    assert not missing, missing
    mod = ast.Module(body=keep, type_ignores=[])
    g = {"np": np}
    if extra_globals:
        g.update(extra_globals)
    exec(compile(mod, path, "exec"), g)
    return g


def load_lines(path, first, last):
    """Source text of 1-based inclusive line range, dedented (for inline script fragments)."""
    import textwrap
    lines = open(path).read().split("\n")[first - 1:last]
    return textwrap.dedent("\n".join(lines))


class InjectedNormal:
    """Stands in for np.random inside of_simulation: normal(scale,size) = scale * next pre-drawn z."""

    def __init__(self, rng):
        self.rng = rng
        self.log = []

    def normal(self, loc=0.0, scale=1.0, size=None):
        z = self.rng.standard_normal(size)
        self.log.append(np.atleast_1d(np.asarray(z, dtype=np.float64)).ravel())
        return loc + scale * z


class NPProxy:
    def __init__(self, random):
        self.random = random

    def __getattr__(self, k):
        return getattr(np, k)


def main():
    sys.dont_write_bytecode = True
    out = {}
    node = load_defs(f"{REF}/velocity_measurment_node", ["generate_test_data", "solve_lgs", "optical_fusion"],
                     {"rospy": None, "time": __import__("time"), "copy": __import__("copy")})
    sim = load_defs(f"{REF}/numerical_simulation/simulation.py",
                    ["generate_test_data", "solve_lgs", "feasibility", "of_simulation"])
    evl = load_defs(f"{REF}/flight_experiments/evaluate_exp.py", ["solve_lgs"])
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, REF)
    import of_library as of_new
    sys.path.pop(0)
    del sys.modules["of_library"]
    sys.path.insert(0, f"{REF}/sensor_precision_experiments/pixhawk_pure_IMU")
    import of_library as of_old
    sys.path.pop(0)
    del sys.modules["of_library"]

    # ---- G1
    pts = np.loadtxt(f"{REF}/numerical_simulation/points.txt")
    v = np.array([1.0, 1, 1]); om = np.array([1.0, 1, 1]); d = 1.0
    n = np.array([0.0, 0, 1]); t = np.array([0.02, 0, 0.205])
    flow = sim["generate_test_data"](pts, v, om, d, n, t)
    vv, R, s = sim["solve_lgs"](pts, flow, d, n, om, t)
    out.update(g1_points=pts, g1_v=v, g1_omega=om, g1_d=d, g1_n=n, g1_t=t, g1_flow=flow,
               g1_v_out=vv, g1_R=np.asarray(R), g1_s=s)
    # node variant on the sim flow (lever-arm identity: returns v + omega x t)
    vn, Rn, rankn, sn = node["solve_lgs"](pts, flow, d, n, om)
    out.update(g1_node_v=vn, g1_node_R=np.asarray(Rn), g1_node_rank=rankn, g1_node_s=sn)

    # ---- G2
    feat = np.array([[-401, 300], [399, -300], [400, 301], [-400, -299]])
    tr = of_new.pix_trans((320, 240))
    x = feat.astype(float).copy()
    x[:, 0] = (x[:, 0] - tr[0]) * 0.01
    x[:, 1] = (x[:, 1] - tr[1]) * 0.01
    u = node["generate_test_data"](x, np.array([1, 1, 1]), np.array([0, 0, 0]), 0.75, np.array([0, 0, 1]))
    v2, R2, rank2, s2 = node["solve_lgs"](x, u, 0.75, np.array([0, 0, 1]), np.array([0, 0, 0]))
    out.update(g2_feat=feat, g2_x=x, g2_u=u, g2_v=v2, g2_R=np.asarray(R2), g2_rank=rank2, g2_s=s2)

    # ---- G3 / G3b / G3c
    r3, d3 = of_new.r_tilde(x, u, np.array([0, 0, 1]), np.array([.1, .1, .1]), .75)
    out.update(g3_r=r3, g3_d=d3)
    rng = np.random.default_rng(33)
    x3 = rng.uniform(-0.6, 0.6, (64, 2)); u3 = rng.normal(0, 0.3, (64, 2))
    u3[5] = 0.0                                    # zero-norm guard (of_library.py:376-378)
    n3 = np.array([0.05, -0.08, 0.99]); n3 /= np.linalg.norm(n3)
    v3 = np.array([0.4, -0.7, 0.2])
    with np.errstate(divide="ignore", invalid="ignore"):
        r3r, d3r = of_new.r_tilde(x3, u3, n3, v3, 1.7)
    out.update(g3r_x=x3, g3r_u=u3, g3r_n=n3, g3r_v=v3, g3r_dist=1.7, g3r_r=r3r, g3r_d=d3r)
    n3neg = -n3                                     # sign-flip branch (of_library.py:380-381)
    with np.errstate(divide="ignore", invalid="ignore"):
        r3n, d3n = of_new.r_tilde(x3, u3, n3neg, v3, 1.7)
    out.update(g3n_r=r3n, g3n_d=d3n)
    fe = sim["feasibility"](pts, v, flow, om, t, n)
    out.update(g3b_out=fe)
    x3h = np.concatenate([x3, np.ones((64, 1))], 1); u3h = np.concatenate([u3, np.zeros((64, 1))], 1)
    keep = np.arange(64) != 5                       # legacy version has no zero guard
    r3c, d3c = of_old.r_tilde(x3h[keep], u3h[keep], n3, v3)
    out.update(g3c_keep=keep, g3c_r=r3c, g3c_d=d3c)

    # ---- G4
    sizes = [3, 4, 20, 200, 500, 2000]
    for k, N in enumerate(sizes):
        rng = np.random.default_rng(100 + k)
        xk = rng.uniform(-0.5, 0.5, (N, 2))
        nk = np.array([rng.normal(0, 0.1), rng.normal(0, 0.1), 1.0]); nk /= np.linalg.norm(nk)
        vk = rng.uniform(-2, 2, 3); ok = rng.normal(0, 0.3, 3); dk = float(rng.uniform(0.5, 5))
        tk = np.array([0.02, 0, 0.205])
        uk = sim["generate_test_data"](xk, vk, ok, dk, nk, tk) + rng.normal(0, 0.01, (N, 2))
        a = node["solve_lgs"](xk, uk, dk, nk, ok)
        b = sim["solve_lgs"](xk, uk, dk, nk, ok, tk)
        c = evl["solve_lgs"](xk, uk, dk, nk, ok, tk)
        p = f"g4_{N}_"
        out.update({p + "x": xk, p + "u": uk, p + "n": nk, p + "v_true": vk, p + "omega": ok, p + "d": dk, p + "t": tk,
                    p + "node_v": a[0], p + "node_R": np.asarray(a[1]), p + "node_rank": a[2], p + "node_s": a[3],
                    p + "sim_v": b[0], p + "sim_R": np.asarray(b[1]), p + "sim_s": b[2],
                    p + "eval_v": c[0], p + "eval_R": np.asarray(c[1])})
    out["g4_sizes"] = np.array(sizes)
    # rank-deficient: every point identical -> A has rank 2, lstsq returns min-norm solution, empty residual
    xd = np.tile(np.array([[0.2, -0.1]]), (5, 1)); ud = np.tile(np.array([[0.3, 0.1]]), (5, 1))
    a = node["solve_lgs"](xd, ud, 1.3, np.array([0, 0, 1.0]), np.array([0.1, 0.0, -0.2]))
    out.update(g4_def_x=xd, g4_def_u=ud, g4_def_v=a[0], g4_def_R=np.asarray(a[1]), g4_def_rank=a[2], g4_def_s=a[3])

    # ---- G5
    S = types.SimpleNamespace
    def msg(secs, nsecs, q, w, a):
        return S(header=S(stamp=S(secs=secs, nsecs=nsecs)),
                 orientation=S(x=q[0], y=q[1], z=q[2], w=q[3]),
                 angular_velocity=S(x=w[0], y=w[1], z=w[2]),
                 angular_velocity_covariance=[1e-3, 0, 0, 0, 2e-3, 0, 0, 0, 3e-3],
                 linear_acceleration=S(x=a[0], y=a[1], z=a[2]))
    q = (0.006070446085061708, -0.004223313821638887, -0.906294856301911, -0.42258129010332524)
    msgs = [msg(100, 937006208, q, [0., 0., 0.], [0.0392266, 0.04903325, 9.40457735]),
            msg(101, 36993408, q, [.1, .2, .3], [0.0392266, 0.0588399, 9.40457735]),
            msg(101, 137000000, (0.1, -0.2, 0.3, 0.9273618495495703), [-.2, .1, .05], [0.3, -0.2, 9.9])]
    st = S(first_imu_=True, got_vel_=False, vel=np.array([.1, .1, .1]))
    call_imu = node["optical_fusion"].call_imu
    states = []
    for m in msgs:
        call_imu(st, m)
        states.append(np.concatenate([st.vel, [st.old_time, st.time_zero], st.rotation.ravel(), st.normal, st.ang, st.ang_err]))
    out["g5_msgs"] = np.array([[m.header.stamp.secs, m.header.stamp.nsecs, m.orientation.x, m.orientation.y,
                                m.orientation.z, m.orientation.w, m.angular_velocity.x, m.angular_velocity.y,
                                m.angular_velocity.z, m.linear_acceleration.x, m.linear_acceleration.y,
                                m.linear_acceleration.z] for m in msgs])
    out["g5_cov_diag"] = np.array([1e-3, 2e-3, 3e-3])
    out["g5_vel0"] = np.array([.1, .1, .1])
    out["g5_states"] = np.array(states)   # vel3, old_time, time_zero, R9, normal3, ang3, ang_err3

    # ---- G6: of_simulation with injected noise (globals: iterations, true_flow, feasibility, solve_lgs)
    sig = dict(ang_vel_sig=0.00071, translation_sig=0.005, height_sig=0.01,
               position_sig=0.056 * 1.23, normal_sig=0.00065)
    for lvl, (fs, ps) in enumerate([(0.0, 0.0), (0.01, np.sqrt(2) / 1000 * 10), (0.056 * np.sqrt(2) * 1.23, 0.056 * 1.23)]):
        inj = InjectedNormal(np.random.default_rng(600 + lvl))
        g = load_defs(f"{REF}/numerical_simulation/simulation.py",
                      ["generate_test_data", "solve_lgs", "feasibility", "of_simulation"])
        g["np"] = NPProxy(inj)
        g["iterations"] = 16
        g["true_flow"] = flow
        vo, feas, Rb = g["of_simulation"](v, om, 1, n, t, pts, sig["ang_vel_sig"], sig["translation_sig"],
                                          sig["height_sig"], fs, ps, sig["normal_sig"])
        out[f"g6_{lvl}_z"] = np.concatenate(inj.log)
        out[f"g6_{lvl}_sig"] = np.array([sig["ang_vel_sig"], sig["translation_sig"], sig["height_sig"], fs, ps, sig["normal_sig"]])
        out[f"g6_{lvl}_v_obs"] = vo
        out[f"g6_{lvl}_bound"] = Rb
        out[f"g6_{lvl}_feasible_last"] = feas

    # ---- G7
    out["g7_in"] = np.array([[320, 240], [480, 640], [321, 241], [1, 1], [1920, 1080]])
    out["g7_out"] = np.array([of_new.pix_trans(tuple(r)) for r in out["g7_in"]], dtype=np.float64)
    rng = np.random.default_rng(7)
    newp = rng.uniform(0, 100, (12, 1, 2)); oldp = newp + rng.normal(0, 2.0, (12, 1, 2))
    oldp[3, 0, 0] = -1.0
    out.update(g7_newpos=newp, g7_oldpos=oldp, g7_static=of_new.static_immobile(newp, oldp, 3.0, 1.5, -1.0))

    # ---- G8
    out["g8_effect_of_flow_errors"] = np.load(f"{REF}/numerical_simulation/effect_of_flow_errors.npy")
    out["g8_effect_of_distance_error"] = np.load(f"{REF}/numerical_simulation/effect_of_distance_error.npy")

    # ---- G9: of_module.py inline system (A_i = [p]x / dist_i, b_i = A_i u_i / (n.p)), lines 139-146
    frag = load_lines(f"{REF}/optical_flow_experiments/of_module.py", 136, 137) + "\n" + \
        load_lines(f"{REF}/optical_flow_experiments/of_module.py", 140, 146).replace("print(A.shape,B.shape)", "pass")
    rng = np.random.default_rng(9)
    fn = np.concatenate([rng.uniform(-0.5, 0.5, (30, 2)), np.ones((30, 1))], 1)
    ff = np.concatenate([rng.normal(0, 0.2, (30, 2)), np.zeros((30, 1))], 1)
    fd = rng.uniform(0.5, 2.0, 30)
    g = {"np": np, "feasible_new": fn, "feasible_flow": ff, "feasible_dist": fd, "n": np.array([0, 0, 1])}
    exec(frag, g)
    out.update(g9_x=fn, g9_u=ff, g9_dist=fd, g9_v=g["v_obs"], g9_R=np.asarray(g["R"]), g9_rank=g["rank"], g9_s=g["s"])

    # ---- G10: node post-solve (lever arm + rotation), line 258 evaluated on G5's final state
    ang = st.ang; offset = np.array([0, 0, 0.1]); v_obs = out["g4_20_node_v"]
    g = {"np": np, "self": S(rotation=st.rotation, ang=ang, offset=offset), "v_obs": v_obs}
    exec(load_lines(f"{REF}/velocity_measurment_node", 258, 258), g)
    out.update(g10_v_obs=v_obs, g10_rotation=st.rotation, g10_ang=ang, g10_offset=offset, g10_v_uav=g["v_uav"])

    np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")
    print("G1 v", out["g1_v_out"], "R", out["g1_R"], "s", out["g1_s"])
    print("G2 v", out["g2_v"], "rank", out["g2_rank"], "s", out["g2_s"])
    print("G3 r", out["g3_r"], "d", out["g3_d"])
    print("G5 vel after #2", out["g5_states"][1][:3])
    print("def rank", out["g4_def_rank"], out["g4_def_v"], out["g4_def_R"], out["g4_def_s"])


if __name__ == "__main__":
    main()
