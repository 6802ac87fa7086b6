#!/usr/bin/env python3
"""Generates tests/golden/jpeg_golden.npz: baseline JPEG byte streams together with the pixels libjpeg-turbo's default
decompressor (ISLOW IDCT, fancy upsampling - the settings cv::imdecode uses for the reference's CompressedImage callback,
velocity_measurment_node.py:112) returns for them.  The library is driven through Pillow, which links the same libjpeg-turbo
and leaves the decompression parameters at their defaults.  Run once in the build container:  python tests/golden/make_golden_jpeg.py
"""
import io
import os

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))


def scene(h, w, seed):
    """A textured colour frame: smooth gradients + blobs + noise (exercises DC prediction, long runs and long codes)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([128 + 90 * np.sin(x / 9.0 + y / 17.0), 128 + 80 * np.cos(x / 13.0 - y / 7.0), 128 + 70 * np.sin((x + 2 * y) / 23.0)], -1)
    for _ in range(12):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, 0.2 * min(h, w))
        img[(x - cx) ** 2 + (y - cy) ** 2 < r * r] = rng.uniform(0, 255, 3)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = [  # name, h, w, mode, subsampling (0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0), quality[, extra encoder options]
    ("c420_odd", 61, 97, "RGB", 2, 80),
    ("c420_ros", 240, 320, "RGB", 2, 80),          # compressed_image_transport's default quality
    ("c420_q100", 128, 160, "RGB", 2, 100),        # long Huffman codes, almost no zero runs
    ("c420_q20", 120, 200, "RGB", 2, 20),          # long zero runs, ZRL symbols
    ("c422", 48, 83, "RGB", 1, 85),
    ("c444", 40, 64, "RGB", 0, 90),
    ("gray", 70, 50, "L", None, 85),
    ("gray_ros", 240, 320, "L", None, 80),
    ("c420_rst_rows", 240, 320, "RGB", 2, 80, {"restart_marker_rows": 1}),       # restart intervals: DRI + RSTn every MCU row
    ("c420_rst_blocks", 61, 97, "RGB", 2, 85, {"restart_marker_blocks": 3}),     # ... every 3 MCUs (RST0..7 wrap around)
    ("c444_rst_blocks", 40, 64, "RGB", 0, 90, {"restart_marker_blocks": 1}),
    ("gray_rst_blocks", 70, 50, "L", None, 85, {"restart_marker_blocks": 5}),
]


def main():
    out = {"libjpeg": np.array(f"libjpeg-turbo {features.version('libjpeg_turbo')} via Pillow"), "names": np.array([c[0] for c in CASES])}
    for i, case in enumerate(CASES):
        name, h, w, mode, ss, q = case[:6]
        img = scene(h, w, 100 + i)
        if mode == "L":
            img = img[:, :, 1]
        buf = io.BytesIO()
        kw = {} if ss is None else {"subsampling": ss}
        kw.update(case[6] if len(case) > 6 else {})
        Image.fromarray(img, mode).save(buf, "JPEG", quality=q, **kw)
        data = buf.getvalue()
        dec = np.asarray(Image.open(io.BytesIO(data)).convert("RGB" if mode == "RGB" else "L"))
        bgr = dec[:, :, ::-1] if mode == "RGB" else np.repeat(dec[:, :, None], 3, axis=2)     # cv2.IMREAD_COLOR layout
        out[f"jpg_{name}"] = np.frombuffer(data, np.uint8)
        out[f"bgr_{name}"] = np.ascontiguousarray(bgr)
    # a progressive stream: must be refused
    buf = io.BytesIO()
    Image.fromarray(scene(32, 32, 7)).save(buf, "JPEG", quality=80, progressive=True)
    out["jpg_progressive"] = np.frombuffer(buf.getvalue(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "jpeg_golden.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.startswith("jpg_")})


if __name__ == "__main__":
    main()
