#!/usr/bin/env python3
"""Cuts the first records out of the reference's recorded sensor logs -> tests/golden/*_excerpt.yaml (data fixtures).

Run in the build container (the reference is not available on the GPU box):  python tests/golden/make_log_excerpts.py
  imuData_excerpt.yaml : first 12 Imu records of optical_flow_experiments/BeispielDatenImuCam22-10-18/imuData.yaml
  hgtData_excerpt.yaml : first 40 Range records of flight_experiments/hgtData.yaml
The files are read as TEXT and split at top-level list items; nothing is parsed or executed here.
"""
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def excerpt(src, dst, n):
    out, count = [], 0
    with open(src, "r") as fh:
        for line in fh:
            if line.startswith("- "):
                count += 1
                if count > n:
                    break
            out.append(line)
    with open(dst, "w") as fh:
        fh.writelines(out)
    print(dst, count - 1 if count > n else count, "records", sum(len(l) for l in out), "bytes")


if __name__ == "__main__":
    excerpt(f"{REF}/optical_flow_experiments/BeispielDatenImuCam22-10-18/imuData.yaml", f"{HERE}/imuData_excerpt.yaml", 12)
    excerpt(f"{REF}/flight_experiments/hgtData.yaml", f"{HERE}/hgtData_excerpt.yaml", 40)
