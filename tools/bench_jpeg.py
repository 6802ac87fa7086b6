#!/usr/bin/env python3
"""Throughput of the compressed ingest (ofk_pairs_upload_jpeg / ofk_jpeg_decode_bgr8): 1080p 4:2:0 quality-80 JPEG streams
(compressed_image_transport's defaults) -> resident BGR frames.  Wall clock around the whole call: host marker parsing, staging
copy, H2D, entropy decode, IDCT, colour conversion.  Beside it: raw ofk_pairs_upload of the same frames (PCIe-bound), the CPU
oracle and, when Pillow is importable, libjpeg-turbo on one host core.

  python tools/bench_jpeg.py [--batch 128] [--reps 5] [--quality 80]
"""
import argparse
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--quality", type=int, default=80)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--chunk", type=int, default=0, help="ofk_set_tuning('jpeg_chunk'): entropy bytes per decoder thread (64, 128, 256; 0 = built-in choice)")
    args = ap.parse_args()
    load_package()
    from of_amd import ofk, synth
    from PIL import Image
    from oracle import jpeg_oracle as jo

    B, H, W = args.batch, args.h, args.w
    if args.chunk:
        ofk.set_tuning("jpeg_chunk", args.chunk)
    distinct = [synth.render_pair(H, W, 900 + k) for k in range(4)]

    def enc(img):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=args.quality, subsampling=2)
        return buf.getvalue()

    sp4 = [enc(p["prev"]) for p in distinct]
    sn4 = [enc(p["next"]) for p in distinct]
    sp = [bytes(bytearray(sp4[b % 4])) for b in range(B)]           # every frame in its own host buffer, as a subscriber sees them
    sn = [bytes(bytearray(sn4[b % 4])) for b in range(B)]
    mean_bytes = float(np.mean([len(s) for s in sp + sn]))
    ctx = ofk.Context(0, W, H, B, 512, 3)
    ctx.pairs_upload_jpeg(sp, sn)                                 # warm-up: scratch + staging allocation
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ctx.pairs_upload_jpeg(sp, sn)
    ctx.sync()
    dt_jpeg = (time.perf_counter() - t0) / args.reps
    t0 = time.perf_counter()
    for k in range(4):
        ctx.jpeg_stage(k & 1, sp + sn)                           # phase 1 alone (host parse + staging copy; returns with the H2D queued)
    dt_stage = (time.perf_counter() - t0) / 4
    ctx._ck(ctx._L.ofk_device_sync())
    st = ctx.jpeg_stage(0, sp + sn); ctx._ck(ctx._L.ofk_device_sync())
    t0 = time.perf_counter()
    ctx.pairs_upload_staged(0, st)                               # phase 2 alone (decoder passes, data already on the device)
    dt_decode = time.perf_counter() - t0
    # double-buffered: a helper thread stages call k + 1 (parse, pinned staging, H2D on the copy stream) while this one decodes call k
    from concurrent.futures import ThreadPoolExecutor
    nb = 2 * args.reps
    with ThreadPoolExecutor(1) as ex:
        fut = ex.submit(ctx.jpeg_stage, 0, sp + sn)
        t0 = time.perf_counter()
        for k in range(nb):
            staged = fut.result()
            fut = ex.submit(ctx.jpeg_stage, (k + 1) & 1, sp + sn) if k + 1 < nb else None
            ctx.pairs_upload_staged(k & 1, staged)
        ctx.sync()
        dt_db = (time.perf_counter() - t0) / nb
    prev = np.stack([distinct[b % 4]["prev"] for b in range(B)])
    nxt = np.stack([distinct[b % 4]["next"] for b in range(B)])
    ctx.pairs_upload(prev, nxt)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ctx.pairs_upload(prev, nxt)
    ctx.sync()
    dt_raw = (time.perf_counter() - t0) / args.reps
    # one decoded frame must equal the oracle's
    ok = bool(np.array_equal(ctx.jpeg_decode([sp4[1]])[0], jo.decode(sp4[1])))
    t0 = time.perf_counter()
    for s in sp4:
        jo.decode(s)
    dt_oracle = (time.perf_counter() - t0) / 4
    t0 = time.perf_counter()
    for s in sp4 * 4:
        np.asarray(Image.open(io.BytesIO(s)).convert("RGB"))
    dt_turbo = (time.perf_counter() - t0) / 16
    print(json.dumps({
        "workload": f"{B} frame pairs {W}x{H}, JPEG 4:2:0 quality {args.quality}, {mean_bytes / 1e3:.0f} kB per frame",
        "jpeg_upload_pairs_per_s": round(B / dt_jpeg, 1), "jpeg_upload_frames_per_s": round(2 * B / dt_jpeg, 1),
        "jpeg_upload_ms_per_call": round(dt_jpeg * 1e3, 2),
        "stage_ms_per_call": round(dt_stage * 1e3, 2), "decode_ms_per_call": round(dt_decode * 1e3, 2),
        "jpeg_double_buffered_pairs_per_s": round(B / dt_db, 1), "jpeg_double_buffered_ms_per_call": round(dt_db * 1e3, 2),
        "raw_upload_pairs_per_s": round(B / dt_raw, 1), "raw_upload_ms_per_call": round(dt_raw * 1e3, 2),
        "cpu_oracle_frames_per_s_1thread": round(1 / dt_oracle, 1), "cpu_libjpeg_turbo_frames_per_s_1thread": round(1 / dt_turbo, 1),
        "matches_oracle": ok}))
    ctx.close()


if __name__ == "__main__":
    main()
