#!/usr/bin/env python3
"""Golden vectors for feas_simulation / overlap (numerical_simulation/simulation.py:70-104, 124-136) from the REFERENCE's own
code, driven exactly like its one live experiment (simulation.py:753-774: three ground planes, the second one with randomly
rotated flow).  Build container only (needs /root/reference); only inputs, the noise that was drawn and the outputs are stored.

    python tests/golden/make_golden_feas.py        # rewrites tests/golden/reference_feas.npz

The functions are AST-extracted (make_golden.load_defs); the constants block (simulation.py:154-178) and the experiment set-up
(:753-754, :762-774) are exec'd by line range.  np.random is replaced by a seeded generator that logs every normal draw in order
(InjectedNormal of make_golden.py, plus uniform for the experiment's rotation angles), so the device kernel and the oracle can
be fed the same noise: per trial 3 + 3 + 1 + 2N + 2N + 3 + 1 + 1 standard normals.
"""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REF, InjectedNormal, NPProxy, load_defs, load_lines  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_feas.npz")
SIM = f"{REF}/numerical_simulation/simulation.py"


class Injected(InjectedNormal):
    def uniform(self, low=0.0, high=1.0, size=None):
        return self.rng.uniform(low, high, size)


def main():
    sys.dont_write_bytecode = True
    warnings.simplefilter("ignore", DeprecationWarning)          # R[i] = one-element array (simulation.py:91)
    out = {}
    for case, (seed, iters) in enumerate([(700, 12), (701, 5)]):
        inj = Injected(np.random.default_rng(seed))
        g = load_defs(SIM, ["generate_test_data", "solve_lgs", "feasibility", "feas_simulation", "overlap"])
        g["np"] = NPProxy(inj)
        g["data"] = np.loadtxt(f"{REF}/numerical_simulation/points.txt")
        exec(load_lines(SIM, 154, 172), g)                       # truth + sigma constants (true_flow of :161 is replaced below)
        g["iterations"] = iters
        if case == 1:                                            # a second case off the experiment's numbers: tilted normal, other sigmas
            g["normal_vector"] = np.array([0.05, -0.03, 1.0]); g["normal_sig"] = 0.02; g["velocity_sig"] = 0.05
        setup = load_lines(SIM, 753, 754) + "\n" + load_lines(SIM, 762, 774)      # centre the points; planes; feas_simulation(...)
        n_before = len(inj.log)
        exec(setup, g)                                           # planes, rotated second plane, feas_simulation(...)
        assert n_before == 0
        N = len(g["data"])
        z = np.concatenate(inj.log)
        assert z.size == iters * (12 + 4 * N), (z.size, iters, N)
        res = [np.asarray(g[k], np.float64) for k in ("backward_para", "backward_dist", "forward_para", "forward_dist", "backward_res", "forward_res")]
        out[f"f{case}_pos"] = g["data"].copy()
        out[f"f{case}_true_flow"] = np.asarray(g["true_flow"], np.float64)
        out[f"f{case}_truth"] = np.concatenate([np.asarray(g["linear_velocity"], np.float64), np.asarray(g["angular_velocity"], np.float64),
                                                [float(g["height_above_gr"])], np.asarray(g["normal_vector"], np.float64),
                                                np.asarray(g["translation"], np.float64), np.asarray(g["linear_velocity"], np.float64)])
        out[f"f{case}_sig"] = np.array([g["ang_vel_sig"], g["translation_sig"], g["height_sig"], g["flow_sig"], g["position_sig"],
                                        g["normal_sig"], g["velocity_sig"]], np.float64)
        out[f"f{case}_z"] = z.reshape(iters, 12 + 4 * N)
        out[f"f{case}_mean"] = np.stack(res)
        # overlap() on the three planes' samples of each statistic, as the histograms of :788-810 group them
        a, b = int(N / 5), 2 * int(N / 3)
        ov = []
        for r in res:
            ov.append([g["overlap"](r[:a], r[a:b]), g["overlap"](r[a:b], r[b:]), g["overlap"](r[:a], r[b:])])
        out[f"f{case}_overlap"] = np.array(ov, np.int64)
        out[f"f{case}_split"] = np.array([a, b])
    # overlap on plain samples, including a degenerate one (all values equal: numpy widens the range by 0.5)
    rng = np.random.default_rng(9)
    d1 = rng.normal(0.0, 1.0, 300); d2 = rng.normal(0.7, 1.5, 257); d3 = np.full(40, 2.5)
    g = load_defs(SIM, ["overlap"])
    out.update(ov_d1=d1, ov_d2=d2, ov_d3=d3, ov_12=g["overlap"](d1, d2), ov_33=g["overlap"](d3, d3), ov_13=g["overlap"](d1, d3))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
