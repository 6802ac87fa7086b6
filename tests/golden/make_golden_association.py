#!/usr/bin/env python3
"""Golden vectors for the sensor association of evaluate_exp.py, produced by the REFERENCE's own lines.

Runs only in the build container (needs /root/reference).  The script fragments are read from the reference file by line
range at generation time and exec'd with prepared inputs (as make_golden.py does for of_module.py); only inputs and the
values those lines produced are written:  python tests/golden/make_golden_association.py -> reference_association.npz

  G11  evaluate_exp.py:68-70, 73-75 (log times relative to the first image) and, per image, :78-80 (nearest IMU / range
       sample by np.argmin(np.abs(...))), :82 (dist), :88-92 (R from the quaternion, normal = R e_z), :95 (omega), on the
       recorded logs of tests/golden/*_excerpt.yaml (parsed by the package's safe loader) and 16 image times.
  G11b the two argmin lines alone on synthetic times with exact ties (first minimum wins).
"""
import os
import sys
import textwrap
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import REF, load_lines  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

SRC = f"{REF}/flight_experiments/evaluate_exp.py"


def main():
    load_package()
    from of_amd import ingest
    S = types.SimpleNamespace
    imu = ingest.load_ros_yaml(os.path.join(HERE, "imuData_excerpt.yaml"))
    height = ingest.load_ros_yaml(os.path.join(HERE, "hgtData_excerpt.yaml"))
    # the two excerpts come from different flights: move the range log onto the IMU log's epoch (its own spacing is kept)
    shift = imu[0].header.stamp.secs - height[0].header.stamp.secs
    for m in height:
        m.header.stamp.secs += shift
    rng = np.random.default_rng(11)
    s0 = imu[0].header.stamp.secs
    cam = np.sort(rng.uniform(0.6, 2.2, 16))
    test = [S(header=S(stamp=S(secs=s0 + int(t), nsecs=int(round((t - int(t)) * 1e9))))) for t in cam]

    g = {"np": np, "imu": imu, "height": height, "test": test}
    exec(load_lines(SRC, 68, 70) + "\n" + load_lines(SRC, 73, 75), g)
    body = "\n".join([load_lines(SRC, 78, 80), load_lines(SRC, 82, 82), load_lines(SRC, 88, 92), load_lines(SRC, 95, 95)])
    rows = []
    g["rows"] = rows
    exec("for i in range(len(test)):\n" + textwrap.indent(body, "    ") +
         "\n    rows.append((current_time, imu_index, hgt_index, dist, R.copy(), normal.copy(), omega.copy()))\n", g)
    out = {
        "g11_secs0": np.array(test[0].header.stamp.secs), "g11_shift": np.array(shift),
        "g11_img_stamps": np.array([[m.header.stamp.secs, m.header.stamp.nsecs] for m in test]),
        "g11_imu_t": g["imu_values"], "g11_hgt_t": g["hgt_values"],
        "g11_t_img": np.array([r[0] for r in rows]), "g11_imu_index": np.array([r[1] for r in rows]),
        "g11_hgt_index": np.array([r[2] for r in rows]), "g11_d": np.array([r[3] for r in rows]),
        "g11_R": np.array([r[4] for r in rows]), "g11_normal": np.array([r[5] for r in rows]),
        "g11_omega": np.array([r[6] for r in rows]),
    }
    # G11b: ties.  times on a 0.25 grid, queries on grid points and exact midpoints
    imu_values = np.array([0.0, 0.5, 0.5, 1.0, 1.5, 1.5, 2.0]); hgt_values = np.array([2.0, 1.0, 0.0, 1.0])
    q = np.array([0.25, 0.5, 0.75, 1.25, 1.5, 1.75, -1.0, 3.0, 1.0, 0.0])
    ii, hi = [], []
    for current_time in q:
        gg = {"np": np, "imu_values": imu_values, "hgt_values": hgt_values, "current_time": current_time}
        exec(load_lines(SRC, 79, 80), gg)
        ii.append(gg["imu_index"]); hi.append(gg["hgt_index"])
    out.update(g11b_imu_t=imu_values, g11b_hgt_t=hgt_values, g11b_q=q, g11b_imu_index=np.array(ii), g11b_hgt_index=np.array(hi))
    # G12: velocity_measurment_node:250-251 (sorted plane distances and their consecutive differences) on the d_i that the
    # reference's own r_tilde returns for the node's test features (G3 of reference_numpy.npz) and on seeded random sets
    node = f"{REF}/velocity_measurment_node"
    g3 = np.load(os.path.join(HERE, "reference_numpy.npz"))
    sets = [np.asarray(g3["g3_d"], np.float64)] + [np.random.default_rng(120 + k).gamma(2.0, 0.7, n) for k, n in enumerate([1, 2, 37, 300])]
    for k, dummy_d in enumerate(sets):
        gg = {"np": np, "dummy_d": dummy_d}
        exec(load_lines(node, 250, 251), gg)
        out[f"g12_{k}_d"] = dummy_d; out[f"g12_{k}_sorted"] = np.asarray(gg["d_sorted"]); out[f"g12_{k}_diff"] = np.asarray(gg["d_diff"], np.float64)
    out["g12_n"] = np.array(len(sets))
    path = os.path.join(HERE, "reference_association.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    print("imu_index", out["g11_imu_index"], "hgt_index", out["g11_hgt_index"])
    print("ties", out["g11b_imu_index"], out["g11b_hgt_index"])


if __name__ == "__main__":
    main()
