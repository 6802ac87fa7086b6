#!/usr/bin/env python3
"""tests/golden/reference_sweeps.npz: the Monte-Carlo sweep outputs the reference holds, as data fixtures.

The eight ''' blocks of numerical_simulation/simulation.py:183-461 each wrote one effect_*.npy (np.append of the per-step mean and
standard deviation of v_obs, 100 steps x 3 components each) from the points of points.txt; the RNG was unseeded, so the files are
statistical pins for k_of_simulation + simulation.sweep (tests/test_gpu_estimation_parity.py).  Nothing of the reference is
executed here: np.load / np.loadtxt of its data files.

  python tests/golden/make_golden_sweeps.py        (build container; /root/reference must be present)"""
import os

import numpy as np

REF = "/root/reference/numerical_simulation"
FILES = ("effect_of_flow_errors", "effect_of_distance_error", "effect_o_ang_vel_error", "effect_of_normal_error",
         "effect_of_translation_error", "effect_of_orientation", "effect_of_height", "effect_of_point_position")


def main():
    out = {"points_raw": np.loadtxt(os.path.join(REF, "points.txt"))}
    for f in FILES:
        a = np.load(os.path.join(REF, f + ".npy"), allow_pickle=False)
        assert a.shape == (600,), (f, a.shape)
        out[f] = a
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_sweeps.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
