#!/usr/bin/env python3
"""Where does the launched run lose its rate?  Times K steps of the resident pipeline (B = 256 x 1080p, two slices) with
(a) nothing behind a step, (b) the per-slice gather of ofk_comm_gather_records, (c) a gather only every 4th step, and prints the
host time spent inside the gather call.  World of one rank (one GPU)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from __graft_entry__ import load_package
load_package()
import of_amd.ofk as ofk
from of_amd import synth, sharding
from of_amd.pipeline import FlowPipeline, PipelineConfig

B, H, W = 256, 1080, 1920
cfg = PipelineConfig.baseline_1080p()
prev, nxt, base = synth.make_batch(B, H, W, seed=2000, distinct=4)
p0 = base[0]
sensors = ofk.make_sensors(B, d=p0["d"], normal=p0["n"], omega=p0["omega"], scaling=p0["scaling"], cx=p0["cx"], cy=p0["cy"])
pipe = FlowPipeline(W, H, B, cfg, streams=2)
pipe.upload(prev, nxt, sensors)
if os.environ.get("WARM_FIRST"):
    pipe.run_async(); pipe.sync()                                # creates the slice / auxiliary streams before RCCL creates its own
ncomm = int(os.environ.get("NCOMM", "2"))
comm = sharding.Comm(pipe.ctx, 0, 1, path="/tmp/ofk_probe_rdv", n_comms=ncomm)

def run(tag, every, K=20):
    for _ in range(3):
        pipe.run_async()
    pipe.sync()
    host = 0.0
    t0 = time.perf_counter()
    for k in range(K):
        pipe.run_async()
        if every and k % every == 0:
            h0 = time.perf_counter(); comm.gather_async(B, k % 2); host += time.perf_counter() - h0
    pipe.sync()
    dt = time.perf_counter() - t0
    print(f"{tag:28s} {B * K / dt:9.0f} pairs/s  {dt / K * 1e3:.3f} ms/step   host time in gather calls {host / K * 1e3:.3f} ms/step", flush=True)

run("no gather", 0)
run("gather every step", 1)
run("gather every 4th step", 4)
run("no gather", 0)
comm.close(); pipe.close()
