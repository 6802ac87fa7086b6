#!/usr/bin/env python3
"""Randomised parity sweep of the device JPEG decoder (GPU box, needs Pillow as the encoder): random sizes, chroma samplings,
qualities (quality 100 on noise = dense 0xFF stuffing, quality 5 = long zero runs), optimised Huffman tables, restart intervals,
batches mixing streams with different tables; every decoded frame compared bit for bit with oracle/jpeg_oracle.
  python tools/stress_jpeg.py [cases] [seed]
"""
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    load_package()
    import of_amd.ofk as ofk
    from PIL import Image
    from oracle import jpeg_oracle as jo
    rng = np.random.default_rng(seed)
    ctx = ofk.Context(0, 1024, 1024, 4, 16, 1)
    bad = 0
    t0 = time.time()
    for case in range(n_cases):
        h, w = int(rng.integers(8, 700)), int(rng.integers(8, 900))
        mode = str(rng.choice(["rgb", "rgb", "rgb", "gray"]))
        ss = None if mode == "gray" else int(rng.choice([0, 1, 2, 2]))
        B = int(rng.integers(1, 5))
        streams = []
        desc = []
        for b in range(B):
            kind = int(rng.integers(0, 4))
            y, x = np.mgrid[0:h, 0:w].astype(np.float64)
            if kind == 0:
                img = rng.integers(0, 256, (h, w, 3)).astype(np.float64)                      # pure noise
            elif kind == 1:
                img = np.stack([128 + 120 * np.sin(x / 5 + b), 128 + 120 * np.cos(y / 7), 128 + 100 * np.sin((x + y) / 11)], -1)
            elif kind == 2:
                img = np.full((h, w, 3), float(rng.integers(0, 256)))                          # flat: EOB-only blocks, 2-3 bits each
                img[:: max(2, h // 5)] = 255
            else:
                img = np.stack([x * 255 / w, y * 255 / h, (x + y) * 255 / (w + h)], -1) + rng.normal(0, float(rng.uniform(0, 40)), (h, w, 3))
            img = np.clip(img, 0, 255).astype(np.uint8)
            if mode == "gray":
                img = np.ascontiguousarray(img[:, :, 0])
            kw = {"quality": int(rng.choice([5, 30, 60, 80, 90, 97, 100]))}
            if ss is not None:
                kw["subsampling"] = ss
            if rng.random() < 0.3:
                kw["optimize"] = True
            r = rng.random()
            if r < 0.25:
                kw["restart_marker_blocks"] = int(rng.integers(1, 40))
            elif r < 0.4:
                kw["restart_marker_rows"] = int(rng.integers(1, 4))
            buf = io.BytesIO()
            try:
                Image.fromarray(img).save(buf, "JPEG", **kw)
            except OSError:                                      # Pillow's output buffer guess fails for some option mixes: plain retry
                kw = {k: v for k, v in kw.items() if k in ("quality", "subsampling")}
                buf = io.BytesIO()
                Image.fromarray(img).save(buf, "JPEG", **kw)
            streams.append(buf.getvalue())
            desc.append(f"k{kind}q{kw['quality']}" + ("o" if kw.get("optimize") else "") + (f"rb{kw['restart_marker_blocks']}" if "restart_marker_blocks" in kw else "")
                        + (f"rr{kw['restart_marker_rows']}" if "restart_marker_rows" in kw else ""))
        try:
            out = ctx.jpeg_decode(streams)
            ok = all(np.array_equal(out[b], jo.decode(streams[b])) for b in range(B))
        except Exception as e:                                   # noqa: BLE001
            ok = False
            print("   exception:", e)
        if not ok:
            bad += 1
        print(f"case {case}: {w}x{h} {mode} ss={ss} B={B} {' '.join(desc)} bytes={[len(s) for s in streams]}: {'ok' if ok else 'MISMATCH'}", flush=True)
    print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
