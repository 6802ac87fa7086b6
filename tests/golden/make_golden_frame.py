#!/usr/bin/env python3
"""tests/golden/reference_frame.npz: the one camera frame the reference holds (flight_experiments/pic2.txt.npy, 240 x 320 x 3 uint8,
BGR as cv_bridge delivered it - it equals picture_test.png in BGR order), read as DATA with numpy.load(allow_pickle=False) and stored
compressed.  Run in the build container (the reference tree does not exist on the GPU box):
    python tests/golden/make_golden_frame.py [/root/reference]"""
import os
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
frame = np.load(os.path.join(ref, "flight_experiments", "pic2.txt.npy"), allow_pickle=False)
assert frame.shape == (240, 320, 3) and frame.dtype == np.uint8
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_frame.npz")
np.savez_compressed(out, frame_bgr=frame, source=np.array("flight_experiments/pic2.txt.npy"))
print(out, os.path.getsize(out), "bytes")
