#!/usr/bin/env python3
"""Where a double-buffered compressed-ingest step spends its wall clock (GPU box):
   python tools/experiments/probe_ingest.py [--batch 512] [--reps 8]
Per batch: time the helper thread needs to stage it (host parse + pinned staging; returns with the H2D queued), time the owner thread
waits for that, time inside ofk_pairs_upload_staged (decoder passes), time to queue the pairs run, and the whole loop."""
import argparse, io, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--reps", type=int, default=8)
    args = ap.parse_args()
    load_package()
    from of_amd import ofk, synth
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    from PIL import Image
    B = args.batch
    ps = [synth.render_pair(1080, 1920, 900 + k) for k in range(4)]

    def enc(img):
        buf = io.BytesIO(); Image.fromarray(img).save(buf, "JPEG", quality=80, subsampling=2); return buf.getvalue()
    sp4 = [enc(p["prev"]) for p in ps]; sn4 = [enc(p["next"]) for p in ps]
    sp = [bytes(bytearray(sp4[b % 4])) for b in range(B)]; sn = [bytes(bytearray(sn4[b % 4])) for b in range(B)]
    pipe = FlowPipeline(1920, 1080, B, PipelineConfig.baseline_1080p())
    sensors = ofk.make_sensors(B, scaling=ps[0]["scaling"], cx=ps[0]["cx"], cy=ps[0]["cy"])
    pipe.upload_jpeg(sp, sn, sensors); pipe.run_async(); pipe.sync()
    t_stage = []

    log = []

    def stage(slot):
        t0 = time.perf_counter(); r = pipe.ctx.jpeg_stage(slot, sp + sn); t_stage.append(time.perf_counter() - t0)
        log.append(("stage", slot, t0, time.perf_counter())); return r
    t_wait, t_dec, t_run = [], [], []
    with ThreadPoolExecutor(1) as ex:
        fut = ex.submit(stage, 0)
        t_all = time.perf_counter()
        for k in range(args.reps):
            t0 = time.perf_counter(); st = fut.result(); t_wait.append(time.perf_counter() - t0)
            fut = ex.submit(stage, (k + 1) & 1) if k + 1 < args.reps else None
            t0 = time.perf_counter(); pipe.ctx.pairs_upload_staged(k & 1, st); t_dec.append(time.perf_counter() - t0)
            log.append(("decode", k & 1, t0, time.perf_counter()))
            t0 = time.perf_counter(); pipe.run_async(); t_run.append(time.perf_counter() - t0)
        pipe.sync()
        t_all = time.perf_counter() - t_all
    base = min(e[2] for e in log)
    for name, slot, a, b in sorted(log, key=lambda e: e[2]):
        print(f"  {name:7s} slot {slot}  {1e3 * (a - base):8.2f} -> {1e3 * (b - base):8.2f} ms")
    ms = lambda v: round(1e3 * float(np.mean(v[1:])), 2)
    print({"batch": B, "stage_ms(helper)": ms(t_stage), "wait_for_stage_ms": ms(t_wait), "decode_ms": ms(t_dec), "queue_run_ms": ms(t_run),
           "loop_ms_per_batch": round(1e3 * t_all / args.reps, 2), "pairs_per_s": round(B * args.reps / t_all, 1)})
    pipe.close()


if __name__ == "__main__":
    main()
