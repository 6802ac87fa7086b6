#!/usr/bin/env python3
"""Per-frame latency of the drop-in node's image callback on the frames it subscribes to (/camerav2_1280x960/image_raw/compressed,
velocity_measurment_node:221): call_optical(CompressedImage) + step() per frame, restored pipeline on the resident stream loop
(JPEG decode on the device, LK from resident tracks, r_tilde filter, solve, re-detection).  Prints one JSON line.
  python tools/bench_node.py [--frames 60] [--w 1280 --h 960]"""
import argparse, io, json, os, sys, time, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60); ap.add_argument("--w", type=int, default=1280); ap.add_argument("--h", type=int, default=960)
    a = ap.parse_args()
    load_package()
    from types import SimpleNamespace
    from PIL import Image
    from of_amd import synth, velocity_node as node
    frames, info = synth.render_sequence(a.h, a.w, 5, 8, v=(0.6, -0.4, 0.2), omega=(0, 0, 0), d=0.75, scaling=0.01)

    def enc(img):
        buf = io.BytesIO(); Image.fromarray(img).save(buf, "JPEG", quality=80, subsampling=2); return buf.getvalue()
    msgs = [SimpleNamespace(format="jpeg", data=enc(f)) for f in frames]
    n = node.optical_fusion(spin=False, synthetic_test=False)
    n.feature_params = dict(qualityLevel=0.05, minDistance=10, blockSize=12); n.T = 2.0
    lat = []
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(a.frames + 4):
            m = msgs[k % len(msgs)] if (k // len(msgs)) % 2 == 0 else msgs[len(msgs) - 1 - k % len(msgs)]      # ping-pong through the clip
            t0 = time.perf_counter(); n.call_optical(m); n.step(); dt = time.perf_counter() - t0
            if k >= 4:
                lat.append(dt)
    lat = np.array(lat) * 1e3
    print(json.dumps({"workload": f"optical_fusion(synthetic_test=False): call_optical(CompressedImage {a.w}x{a.h}, JPEG q80 4:2:0, "
                                  f"{np.mean([len(m.data) for m in msgs]) / 1e3:.0f} kB) + step() per frame, resident stream loop",
                      "frames": len(lat), "ms_per_frame_mean": round(float(lat.mean()), 3), "ms_per_frame_p50": round(float(np.median(lat)), 3),
                      "ms_per_frame_p95": round(float(np.percentile(lat, 95)), 3), "tracks": int(len(n.feat)),
                      "camera_period_ms_at_20Hz": 50.0}))


if __name__ == "__main__":
    main()
