#!/usr/bin/env python3
"""bench.py — frame-pairs/s of the optical-flow -> ego-velocity hot path on N MI355X GPUs.

  python bench.py [--gpus N --steps K --warmup W --batch B] [--config c1|c2|c4]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Default (--config c1) = BASELINE.json configs[1], the configuration the metric is quoted on: 1920x1080 frame pairs, 500 Shi-Tomasi
corners, 3-level LK pyramid.  --config c2 = configs[2] (1024 independent 640x480 pairs per step + the per-pair 6-state filter update
queued behind every step), --config c4 = configs[4]'s pair workload (3840x2160, 2000 corners, 5-level pyramid, 256 pairs per step);
same JSON contract, `config.workload` names the BASELINE entry, roofline and traffic come from that configuration's own PMC passes
(profiles/<tag>c2_*, <tag>c4_*).

A "step" is one pass of the whole hot path (BGR pair -> gray -> pyramids -> corners -> LK -> velocity) over a batch
of B synthetic frame pairs per GPU that are already resident in HBM.  Frame pairs are independent, so ranks
shard them with no data-path collective ("weak" scaling: B pairs per GPU); the only exchange is an RCCL all_gather
of the [B,8] f32 velocity records after every step.  Timing: W warm-up steps, then exactly K steps between
barrier + device synchronize on both sides, MAX over ranks.  torch.distributed.run is only the process LAUNCHER here
(it exports RANK / LOCAL_RANK / WORLD_SIZE): this file never imports torch.  Barrier, MAX and the all_gather are RCCL
collectives issued by libofk.so itself (ofk_comm_*: librccl.so bound with dlopen, unique id handed over through a
node-local file, of_amd/sharding.py), hipDeviceSynchronize through the library stands in for torch.cuda.synchronize().

The gather of step k is queued on the library's own stream right behind step k's solve (stream-ordered, no host wait), so
step k+1 is launched while it travels; K steps issue K gathers inside the timed region.

Schedule (default): one stage chain (response -> selection -> LK -> solve) on the context's stream, the HBM-bound gray
conversions and pyramids of the NEXT step on an auxiliary stream beside it; B = 512 pairs per step at 1080p.
c2 runs two free-running slices (at 480p LK is the longest stage and the latency-bound phases weigh more: +10 %).

The JSON line also carries
  roofline     : the dominant kernel = the longest stage of the step's critical chain (response -> select -> LK -> solve).
                 Its duration is taken from the SERIAL pass that follows the timed region (same process, same resident
                 frames, every stage alone on one stream, HIP events on that stream): under the overlapped schedule
                 kernels share the chip and an event bracket then measures contention, not the kernel.  The dominant
                 kernel is bound by VALU issue, so `bound` is "valu": achieved = wave-level VALU instructions per launch
                 (committed SQ_INSTS_VALU count, profiles/<tag>_valu_pmc.json) / that duration, peak = one wave instruction per
                 2 clocks per SIMD at 2.4 GHz on 1024 SIMDs (MI355X_MICROARCH.md) = 1228.8 G/s; `isa_mix` prices the same
                 kernel against the issue cycles of its own instruction mix (profiles/<tag>_isa_mix.json), once at 2.4 GHz and
                 once at the rate the chip sustains (v_add_u32 back to back on every SIMD: tools/valu_rates.hip,
                 profiles/<tag>_valu_rates.txt); `hbm` keeps the HBM view (SURVEY.md §8(d) algorithmic bytes per launch / the same
                 duration) and `traffic` the HBM bytes per launch from the committed PMC passes (profiles/<tag>_traffic_pmc.json).
  north_star_group : SURVEY.md §8(d)'s "pyramid + LK group" = (G_pyr + G_lk) x B / (t_pyr + t_lk), serial-pass times.
  north_star_kernel: the pyramid kernel alone (the HBM-bound member of that group) with THREE fractions of the 8 TB/s peak:
                 `frac` by SURVEY's G_pyr (which counts re-reading levels 1 and 2), `frac_fused_minimum` by the bytes a one-pass
                 kernel must move (2 x (P0 + .. + PL)), `frac_traffic` by the PMC bytes.
  stages       : every stage under the default overlapped schedule (event brackets from K more steps right after the timed region;
                 they include contention from the other streams and are NOT kernel durations).
  stages_isolated : every stage alone on the chip (the serial pass), with its HBM fraction and its PMC traffic; a stage whose
                 traffic lies BELOW its algorithmic bytes is flagged: SURVEY's byte count is then not that kernel's minimum.
  ingest_inclusive : pairs/s when the frames are NOT resident: raw BGR over PCIe every step, and JPEG streams decoded on the
                 device every step (bounded: a few steps each, outside the timed region) with the decoder's own kernel times;
                 jpeg_double_buffered.value = 40 batches with the first one's staging unhidden, .steady_state = the loop's period.
  cpu_baseline : the CPU oracle (oracle/, single thread, kind "port") timed on this host over a bounded sample of the
                 same frame pairs (rank 0, N=1 only).
  monte_carlo_sweep (--config c4 only): configs[4]'s Monte-Carlo sweep at 2000 points x 4096 trials per step, device noise and host noise.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before anything initialises the HIP runtime (torch does, below): 4 pipeline streams + torch's + RCCL's need more than the
# runtime's default 4 hardware queues, or two of them share a queue and serialise (schedule)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from __graft_entry__ import load_package  # noqa: E402


class _stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block (C stdio of loaded libraries included): stdout stays ONE JSON line."""

    def __enter__(self):
        import ctypes
        self._libc = ctypes.CDLL(None)
        sys.stdout.flush(); self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush(); self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2.0      # 1228.8 G wave-instr/s: wave64 VALU = 2 clocks on a SIMD-32, 1024 SIMDs, 2.4 GHz
PROFILE_TAG = "r04"          # profiles/<tag>_traffic_pmc.json, _valu_pmc.json, _isa_mix.json (tools/profile_round.sh); <tag>c2_*, <tag>c4_* for the other configs
KERNEL_OF = {"gray": "k_gray_bgr8", "pyr": "k_pyr3_stream", "eig": "k_mineig_pair<7,false>", "select": "k_select_greedy", "lk": "k_lk15q",
             "solve": "k_pairs_solve", "nms": "-"}
# BASELINE.json configs the bench can run on one GPU (configs[0] is the CPU plumbing case, configs[3] needs eight GPUs' cameras:
# tests/test_gpu_baseline_full.py; tools/bench_configs.py times its per-frame loop)
CONFIGS = {
    "c1": dict(h=1080, w=1920, corners=500, levels=3, batch=512, streams=1, ekf=False, cpu_sample=24, tag="",
               workload="1920x1080 frame pairs, 500 Shi-Tomasi corners, 3-level LK pyramid (BASELINE configs[1])"),
    "c2": dict(h=480, w=640, corners=500, levels=3, batch=1024, streams=2, ekf=True, cpu_sample=128, tag="c2",
               workload="batch of 1024 independent 640x480 frame pairs + per-pair 6-state EKF update (BASELINE configs[2])"),
    "c4": dict(h=2160, w=3840, corners=2000, levels=5, batch=256, streams=1, ekf=False, cpu_sample=8, tag="c4",
               workload="3840x2160 frame pairs, 2000 corners, 5-level pyramid: the pair workload of BASELINE configs[4] "
                        "(its Monte-Carlo sweep: simulation.sweep, tests/test_gpu_estimation_parity.py)"),
}
CFG_TAG = ""                 # set by main(): "" / "c2" / "c4" - which configuration's profile summaries apply


def algorithmic_bytes(cfg, n_pts, n_cand, H, W):
    """algorithmic bytes per frame pair, per stage (L levels, N points, w window)."""
    P = [H * W]
    h, w = H, W
    for _ in range(cfg.max_level):
        h, w = (h + 1) // 2, (w + 1) // 2
        P.append(h * w)
    L = cfg.max_level
    wn = cfg.win
    return {
        "gray": 2 * (3 * P[0] + P[0]),
        "pyr": 2 * sum(P[l - 1] + P[l] for l in range(1, L + 1)),
        # what a pass that builds every level from registers must move: level 0 in, levels 1..L out (both frames)
        "pyr_fused_minimum": 2 * sum(P),
        # SURVEY's G_eig (P + 4P) and G_nms (4P + 8 N_cand) assume a materialised f32 response map.  The fused streaming
        # kernel never writes it: its compulsory traffic is the gray frame in and the candidate keys out.
        "eig": P[0] + 8 * n_cand,
        "nms": 0,
        "select": 8 * n_cand + 8 * n_pts,
        "lk": n_pts * sum((wn + 2) ** 2 + (wn + 1 + 2 * 3) ** 2 for _ in range(L + 1)) + 21 * n_pts,
        "solve": n_pts * 32 + 200,
    }


def _profile(name):
    """The committed summary `name` of the configuration being run: this round's, else (configs[1] only) the latest earlier one."""
    tags = [PROFILE_TAG + CFG_TAG] + ([] if CFG_TAG else ["r03", "r02", "r01"])
    for tag in tags:
        try:
            return json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{name}.json"))), tag
        except Exception:
            continue
    return None, None


def pmc_traffic(stage, pairs_per_launch):
    """HBM bytes per launch group of the stage from the committed rocprofv3 PMC passes (profiles/<tag>_traffic_pmc.json,
    written by tools/profile_round.sh + tools/make_traffic_json.py from this same command), scaled to the pairs one launch
    group processes; None if absent."""
    t, _ = _profile("traffic_pmc")
    try:
        return int(t["stages"][stage]["hbm_bytes_per_step"] * pairs_per_launch / t["batch"])
    except Exception:
        return None


def checked_traffic(stage, pairs_per_launch, algorithmic_bytes_per_launch):
    """pmc_traffic, refused (None + a note on stderr) when it lies below 0.8 x the kernel's algorithmic bytes: a cold kernel cannot
    move less than it must, so such a number is a bookkeeping error (round 2's summary halved every stage)."""
    t = pmc_traffic(stage, pairs_per_launch)
    if t is not None and t < 0.8 * algorithmic_bytes_per_launch:
        print(f"bench.py: PMC traffic of stage {stage} ({t} B per launch) is below 0.8 x its algorithmic bytes ({int(algorithmic_bytes_per_launch)} B): "
              f"refused, reporting null (regenerate profiles/{PROFILE_TAG}{CFG_TAG}_traffic_pmc.json with tools/profile_round.sh)", file=sys.stderr)
        return None
    return t


def valu_roof(stage, pairs_per_launch, isolated_ms):
    """The stage's kernel against the VALU issue peak: wave-level instructions per launch (committed SQ_INSTS_VALU count, scaled
    to the pairs of one launch) / the kernel's duration alone on the chip.  Returns (achieved G/s, instructions per launch,
    source file tag) or None."""
    t, tag = _profile("valu_pmc")
    try:
        n = t["stages"][stage]["SQ_INSTS_VALU_per_launch"] * pairs_per_launch / t["batch"]
        return n / (isolated_ms * 1e-3) / 1e9, int(n), tag
    except Exception:
        return None


def isa_mix_roof(stage, pairs_per_launch, isolated_ms):
    """Issue-cycle bound of the kernel's own instruction mix (tools/isa_mix.py -> profiles/<tag>_isa_mix.json: VALU issue cycles
    per pair = sum over instruction classes of count x measured issue cost): the time the chip's 1024 SIMDs need to issue that
    mix at 2.4 GHz with no stall at all, over the measured duration."""
    t, tag = _profile("isa_mix")
    try:
        cyc = t["stages"][stage]["valu_issue_cycles_per_pair"] * pairs_per_launch
        floor_ms = cyc / (1024 * 2.4e9) * 1e3
        out = {"issue_floor_ms": round(floor_ms, 4), "frac": round(floor_ms / isolated_ms, 4),
               "mean_cycles_per_valu_instr": t["stages"][stage].get("mean_cycles_per_valu_instr"), "source": f"profiles/{tag}_isa_mix.json"}
        # the same against the rate the chip SUSTAINS: tools/valu_rates.hip measures v_add_u32 (2 issue clocks) back to back on every
        # SIMD at 0.95 ns per wave-instruction per SIMD, i.e. 1.05 T/s chip-wide where 2.4 GHz would give 1.2288 T/s
        rates = os.path.join(ROOT, "profiles", f"{tag}_valu_rates.txt")
        if os.path.exists(rates):
            import re
            m = re.search(r"^v_add_u32\s+[\d.]+ ms\s+([\d.]+) ns/instr/SIMD", open(rates).read(), re.M)
            if m:
                ns_per_clock = float(m.group(1)) / 2.0
                sustained_ms = cyc / 1024 * ns_per_clock * 1e-6
                out.update({"issue_floor_ms_at_measured_rate": round(sustained_ms, 4), "frac_at_measured_rate": round(sustained_ms / isolated_ms, 4),
                            "measured_full_rate_ns_per_wave_instr_per_simd": float(m.group(1)), "rates_source": f"profiles/{tag}_valu_rates.txt"})
        return out
    except Exception:
        return None


def ingest_inclusive(pipe, prev, nxt, sensors, B, reps=2):
    """Pairs/s when the frames are not resident (never `value`): every step uploads its B pairs first — as raw BGR over PCIe
    (ofk_pairs_upload, 12.4 MB per 1080p pair) and as baseline JPEG streams decoded on the device (ofk_pairs_upload_jpeg; what
    the reference's node receives, node:112,221).  Wall clock around upload + ofk_pairs_run + sync, `reps` steps each."""
    out = {}
    try:
        pipe.upload(prev, nxt, sensors); pipe.run_async(); pipe.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            pipe.upload(prev, nxt, sensors); pipe.run_async()
        pipe.sync()
        dt = time.perf_counter() - t0
        out["raw_bgr_upload"] = {"value": round(B * reps / dt, 1), "unit": "frame-pairs/s", "bytes_per_pair": int(prev[0].nbytes + nxt[0].nbytes)}
    except Exception as e:                                       # never lose the bench line over an optional leg
        out["raw_bgr_upload"] = {"error": str(e)[:200]}
    try:
        import io as _io
        from PIL import Image

        def enc(img):
            buf = _io.BytesIO()
            Image.fromarray(img).save(buf, "JPEG", quality=80, subsampling=2)      # compressed_image_transport's defaults
            return buf.getvalue()
        D = min(8, B)
        sp = [enc(prev[b]) for b in range(D)]; sn = [enc(nxt[b]) for b in range(D)]
        jp = [bytes(bytearray(sp[b % D])) for b in range(B)]; jn = [bytes(bytearray(sn[b % D])) for b in range(B)]
        pipe.upload_jpeg(jp, jn, sensors); pipe.run_async(); pipe.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            pipe.upload_jpeg(jp, jn, sensors); pipe.run_async()
        pipe.sync()
        dt = time.perf_counter() - t0
        out["jpeg_decode_on_device"] = {"value": round(B * reps / dt, 1), "unit": "frame-pairs/s",
                                        "bytes_per_pair": int(np.mean([len(a) + len(b) for a, b in zip(jp, jn)])),
                                        "streams": f"{D} distinct 1080p 4:2:0 quality-80 frames per side, repeated"}
        # the same with the ingest double-buffered (FlowPipeline.run_jpeg_batches): a helper thread parses, stages and uploads batch
        # k + 1 while the GPU decodes batch k and runs its pairs
        nb = 40                                                  # 20 k pairs; the first batch stages unhidden: `steady_state` below leaves the ramp out
        pipe.run_jpeg_batches([(jp, jn)] * 2, sensors); pipe.sync()        # warm-up: both staging slots allocate their pinned / device buffers
        marks = []
        t0 = time.perf_counter()
        pipe.run_jpeg_batches([(jp, jn)] * nb, sensors, on_step=lambda k: marks.append(time.perf_counter()))
        pipe.sync()
        dt = time.perf_counter() - t0
        # `value`: all nb batches, the first one's staging not hidden (round 3's definition); `steady_state`: the period between queued
        # steps once the loop is full (batches 3 .. nb)
        steady = (marks[-1] - marks[2]) / (len(marks) - 3) if len(marks) > 4 else None
        out["jpeg_double_buffered"] = {"value": round(B * nb / dt, 1), "unit": "frame-pairs/s", "batches": nb,
                                       "steady_state": round(B / steady, 1) if steady else None,
                                       "how": "ofk_jpeg_stage of batch k+1 on a helper thread (host parse, pinned staging, H2D on the copy stream) "
                                              "while ofk_pairs_upload_staged decodes batch k; both frames of a pair in one decoder batch"}
    except Exception as e:
        out["jpeg_decode_on_device"] = {"error": str(e)[:200]}
    # the decoder's own kernels (committed counter passes: tools/experiments/pmc_jpeg.sh -> profiles/<tag>_jpeg_decoder.json): per
    # 512 frames of 1080p, every kernel alone on the chip, the dominant one against the VALU issue peak
    dec, tag = _profile("jpeg_decoder")
    if dec:
        try:
            ks = {k: v for k, v in dec["kernels"].items() if k.startswith("k_jpeg")}
            dom = max(ks, key=lambda k: ks[k]["ms_per_512_frames"])
            out["decoder"] = {"source": f"profiles/{tag}_jpeg_decoder.json", "ms_per_512_frames": dec["decoder_ms_per_512_frames"],
                              "kernels_ms_per_512_frames": {k: v["ms_per_512_frames"] for k, v in ks.items()},
                              "dominant": {"kernel": dom, "launches_per_decode": ks[dom]["launches_per_decode"], "ms_per_512_frames": ks[dom]["ms_per_512_frames"],
                                           "bound": "valu", "achieved": ks[dom]["valu_ginstr_per_s"], "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
                                           "frac": ks[dom]["valu_frac_of_1228.8"], "wait_share_of_wave_cycles": ks[dom]["wait_share_of_wave_cycles"]}}
        except Exception as e:
            out["decoder"] = {"error": str(e)[:200]}
    return out


def monte_carlo_sweep(points=2000, trials=4096, steps=10):
    """The other half of BASELINE configs[4]: the reference's effect_of_flow_errors sweep (simulation.py:183-202) at 2000 points and
    4096 trials per sigma step - of_simulation for all trials of a step in ONE launch, the normals drawn on the device by the
    counter-based generator (ofk_of_simulation_rng), against the same sweep with numpy's generator and a 262 MB noise tensor uploaded
    per step.  Bounded: `steps` sigma steps each way; never part of `value`."""
    import of_amd.simulation as sim
    out = {"workload": f"effect_of_flow_errors sweep, {points} points x {trials} trials per sigma step, f64", "unit": "trials/s"}
    try:
        pts = np.random.default_rng(1).uniform(-1.2, 1.2, (points, 2))
        sargs = (pts, [1.0, 1, 1], [1.0, 1, 1], 1.0, [0.0, 0, 1], [0.02, 0, 0.205])
        for name, kw, n in (("device_noise", dict(device_seed=7), steps), ("host_noise", dict(generator=np.random.default_rng(3)), max(2, steps // 4))):
            sim.sweep("flow_errors", *sargs, k=100, trials=trials, steps=[5], **kw)
            t0 = time.perf_counter()
            res = sim.sweep("flow_errors", *sargs, k=100, trials=trials, steps=list(range(10, 10 + n)), **kw)
            dt = (time.perf_counter() - t0) / n
            out[name] = {"value": round(trials / dt, 1), "ms_per_sigma_step": round(dt * 1e3, 3), "v_mean_of_first_step": [round(float(x), 5) for x in res[:3]]}
    except Exception as e:
        out["error"] = str(e)[:200]
    return out


def cpu_baseline(prev, nxt, sensors, cfg, sample, threads=1):
    """CPU oracle over `sample` pairs of the same workload on `threads` host threads (the C image stages release the GIL);
    test infrastructure used as the checker's timing leg only.  Returns (pairs/s, velocity of the last pair)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import image_oracle as io, estimation_oracle as eo
    io.lib()

    def one(b):
        g0, g1 = io.gray_bgr8(prev[b]), io.gray_bgr8(nxt[b])
        pts = io.good_features(g0, cfg.max_corners, cfg.quality, cfg.min_distance, cfg.block_size)
        n, s, e = io.lk_pyr(g0, g1, pts, cfg.win, cfg.max_level, cfg.max_count, cfg.eps, cfg.min_eig_thr)
        ok = s.ravel() == 1
        sr = sensors[b]
        new = n.reshape(-1, 2).astype(np.float64); old = pts.reshape(-1, 2).astype(np.float64)
        x = (new[ok] - [sr[20], sr[21]]) * sr[19]; u = (new[ok] - old[ok]) * sr[19]
        return eo.solve_lgs_node(x, u, sr[0], sr[1:4], sr[4:7])[0]

    t0 = time.perf_counter()
    if threads <= 1:
        v = [one(b) for b in range(sample)]
    else:
        with ThreadPoolExecutor(threads) as ex:
            v = list(ex.map(one, range(sample)))
    dt = time.perf_counter() - t0
    return sample / dt, v[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c1", choices=sorted(CONFIGS) + ["1", "2", "4"], help="BASELINE.json configuration: c1 = configs[1] (default, the "
                    "metric's own: 1080p / 500 corners / 3 levels), c2 = configs[2] (1024 x 640x480 + per-pair EKF), c4 = configs[4]'s 4K pair workload")
    ap.add_argument("--batch", type=int, default=0, help="frame pairs per GPU per step (default: 512 at 1080p = 14 GB of the 288 GB; 1024 for c2; 256 for c4)")
    ap.add_argument("--streams", type=int, default=0, help="free-running slices of the batch (HIP streams) per GPU (default: 1; 2 for c2)")
    ap.add_argument("--no-overlap", action="store_true", help="run every stage of a step serially on one stream")
    ap.add_argument("--no-isolated", action="store_true", help="skip the serial pass behind the timed region (profiling: the kernel "
                    "trace then holds the timed schedule's launches only)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="pairs timed on the CPU oracle (0 = skip; default 24 at 1080p, 128 for c2, 8 for c4)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the ingest-inclusive legs (raw BGR upload / JPEG decode every step)")
    ap.add_argument("--comms", type=int, default=1, help="RCCL communicators per rank under a launcher: 1 (default) = one gather per step behind the "
                    "last slice - the plain usage every RCCL build supports; S = --streams: one communicator per slice, every slice gathers its own "
                    "records on its own stream (+0.8 %% at N = 1; two collectives of one rank then run concurrently)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE", help="ofk_set_tuning knob for experiments (eig_rows, no_pair, "
                    "no_pyr3, pyr3_chunks, pyr_rows, jpeg_chunk); results do not depend on them")
    ap.add_argument("--watchdog", type=float, default=90.0, help="seconds the bootstrap + first gathered step may take under a launcher before "
                    "the rank reports what it is waiting for and exits 3 (0 = off)")
    args = ap.parse_args()
    global CFG_TAG
    cname = args.config if args.config in CONFIGS else "c" + args.config
    C = CONFIGS[cname]
    CFG_TAG = C["tag"]
    H, W = C["h"], C["w"]
    args.batch = args.batch or C["batch"]
    args.streams = args.streams or C["streams"]                  # = pipeline.auto_streams(w, h, batch): two slices up to 640x480, else one
    if args.cpu_sample < 0:
        args.cpu_sample = C["cpu_sample"]

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # RCCL's device-memory IPC needs the dmabuf path on this pool
    load_package()
    import of_amd.ofk as ofk
    from of_amd import synth, sharding
    from of_amd.pipeline import FlowPipeline, PipelineConfig
    for kv in args.tune:
        k, v = kv.split("=")
        ofk.set_tuning(k, int(v))
    rank, world, local = sharding.env_ranks()                   # RANK / WORLD_SIZE / LOCAL_RANK from the launcher; (0, 1, 0) when run plainly
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; reporting n_gpus={world}", file=sys.stderr)

    cfg = PipelineConfig(max_corners=C["corners"], quality=0.01, min_distance=10, block_size=7, win=15, max_level=C["levels"], max_count=20, eps=0.03)
    B = args.batch
    truth = dict(v=(0.002, -0.0015, 0.001), omega=(0.002, -0.001, 0.003), d=1.0)
    prev, nxt, base = synth.make_batch(B, H, W, seed=2000 + 131 * rank, distinct=4, **truth)
    p0 = base[0]
    sensors = ofk.make_sensors(B, d=p0["d"], normal=p0["n"], omega=p0["omega"], scaling=p0["scaling"], cx=p0["cx"], cy=p0["cy"])
    pipe = FlowPipeline(W, H, B, cfg, device=local, streams=args.streams)
    if args.no_overlap:
        pipe.ctx.set_overlap(False)
    pipe.upload(prev, nxt, sensors)
    if C["ekf"]:                                                 # configs[2]: the 6-state filter of every pair, state resident, updated behind each step
        from of_amd.pipeline import FilterModel
        pipe.ctx.filter_configure(FilterModel.ekf6(), B)

    # The exchange: RCCL through the library (no torch in the process).  Step k's records are exported and all-gathered on the
    # library's own stream right behind step k's solve (receive slot k % 2), so the host only queues work: step k+1 is
    # launched while step k's gather travels.  K steps issue K gathers inside the timed region.
    # Watchdog for the first multi-rank step: a rank that waits longer than --watchdog seconds for the bootstrap or the first
    # gathered step says which rank / communicator / slot it is waiting for and exits non-zero (no retry, no re-exec).
    state = {"phase": "bootstrap (unique-id file + ncclCommInitRank)", "comm": None}
    armed = None
    if launched and args.watchdog > 0:
        import threading
        armed = threading.Event()

        def bark():
            if armed.wait(args.watchdog):
                return
            c = state["comm"]
            pend = [c.pending(s) for s in (0, 1)] if c is not None else None
            print(f"bench.py watchdog: rank {rank} of {world} (device {local}) has been in '{state['phase']}' for {args.watchdog:.0f} s; "
                  f"communicators agreed: {c.n_comms if c is not None else 'none yet'}; gathers still travelling per slot (bit k = slice k): {pend}; "
                  f"rendezvous file: {sharding.rendezvous_path(generation=0)}; try --comms 1", file=sys.stderr, flush=True)
            os._exit(3)
        threading.Thread(target=bark, daemon=True).start()
    n_comms = max(1, min(args.comms, args.streams))
    with _stdout_to_stderr():                                    # librccl prints a version banner on stdout at communicator init
        comm = sharding.Comm(pipe.ctx, rank, world, n_comms=n_comms) if launched else None
    state["comm"] = comm
    state["phase"] = "warm-up: first steps + gathers + barrier"

    def sync_all():
        if comm is not None:
            comm.barrier()
        pipe.ctx._ck(pipe.ctx._L.ofk_device_sync())

    def run_steps(n, k0):
        for k in range(k0, k0 + n):
            pipe.run_async()
            if C["ekf"]:
                pipe.ctx.pairs_filter_step(B)
            if comm is not None:
                comm.gather_async(B, k % 2)

    run_steps(args.warmup, 0)
    sync_all()
    if armed is not None:
        armed.set()                                              # the first gathered steps and a barrier completed on every rank
    pipe.ctx.profile_read()
    pipe.ctx.profile_enable(0)                                   # the timed region runs without stage brackets (two event records per stage and slice)
    sync_all()
    t0 = time.perf_counter()
    run_steps(args.steps, args.warmup)
    sync_all()
    dt = time.perf_counter() - t0
    # the same K steps once more under the same schedule with every stage bracketed by events: the `stages` object (outside the timed region)
    pipe.ctx.profile_enable(0x7f)
    run_steps(args.steps, args.warmup + args.steps)
    sync_all()
    prof = pipe.ctx.profile_read()
    pipe.ctx.profile_enable(0)
    # Outside the timed region: the same K steps once more with every stage serial on one stream, so that each kernel's
    # duration is its own (under overlap two kernels share the chip and each one's event time includes the other).
    iso = prof
    if (not args.no_overlap or args.streams > 1) and not args.no_isolated:
        pipe.ctx.set_overlap(False)
        pipe.ctx.set_streams(1)
        pipe.run_async(); pipe.sync()
        pipe.ctx.profile_enable(0x7f)
        for _ in range(args.steps):
            pipe.run_async()
        pipe.sync()
        iso = pipe.ctx.profile_read()
        pipe.ctx.profile_enable(0)
        pipe.ctx.set_overlap(not args.no_overlap)
        pipe.ctx.set_streams(args.streams)

    if comm is not None:
        dt = comm.max(dt)
        last = (args.warmup + 2 * args.steps - 1) % 2
        gathered = comm.fetch(B, last)                             # [world, B, 8] f32 of the last gathered step, every rank's records

    out = pipe.ctx.pairs_download(points=False)
    rec = out["records"]
    if comm is not None:                                         # the gather carried this rank's own records, bit for bit (f32)
        mine = np.stack([rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 11], rec[:, 7], rec[:, 4], rec[:, 12]], 1).astype(np.float32)
        if not np.array_equal(gathered[rank], mine) or not np.all(np.isfinite(gathered)):
            raise SystemExit(f"rank {rank}: gathered records differ from the local ones")
    if rank == 0:
        n_pts = float(np.mean(out["counts"]))
        n_cand = float(np.mean(rec[:, 14]))
        ab = algorithmic_bytes(cfg, n_pts, n_cand, H, W)
        stages = {}
        for s in ofk.STAGES:
            ms, nl = prof[s]
            per_step = ms / args.steps                   # summed over the slices (streams) of a step
            stages[s] = {"ms_per_step": round(per_step, 4), "launches_per_step": nl // max(1, args.steps),
                         "algorithmic_GBps": round(ab[s] * B / (per_step * 1e-3) / 1e9, 2) if per_step > 0 else None}
        # dominant kernel = the longest stage of the critical chain, by its duration ALONE on the chip (serial pass): event
        # brackets under the overlapped schedule include the other slice's kernels and are not kernel durations
        chain = [s for s in ofk.STAGES if s not in ("gray", "pyr", "nms")]
        iso_ms = {s: (iso[s][0] / iso[s][1] if iso[s][1] else 0.0) for s in ofk.STAGES}     # mean duration of one launch (B pairs)
        iso_pairs = B                                             # the serial pass runs one slice: every launch covers the whole batch
        dom = max(chain, key=lambda s: iso_ms[s])
        dom_ms = iso_ms[dom]
        hbm_ach = ab[dom] * iso_pairs / (dom_ms * 1e-3) / 1e9
        vr = valu_roof(dom, iso_pairs, dom_ms)
        roof = {"kernel": f"{KERNEL_OF[dom]} (stage {dom})", "avg_ms": round(dom_ms, 4), "pairs_per_launch": iso_pairs,
                "timing": "mean HIP-event duration of the kernel's launches in the serial pass of this run (every stage alone on one "
                          "stream, after the timed region); a kernel's time per step is <= ms_per_step by construction",
                "hbm": {"achieved": round(hbm_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_ach / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": int(ab[dom] * iso_pairs)},
                "traffic": checked_traffic(dom, iso_pairs, ab[dom] * iso_pairs)}
        if vr is not None:
            roof.update({"bound": "valu", "achieved": round(vr[0], 1), "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s",
                         "frac": round(vr[0] / VALU_PEAK_GINSTR, 4), "valu_instr_per_launch": vr[1],
                         "valu_source": f"profiles/{vr[2]}_valu_pmc.json", "isa_mix": isa_mix_roof(dom, iso_pairs, dom_ms)})
        else:                                                     # no committed instruction count: only the HBM view can be stated
            roof.update({"bound": "hbm", "achieved": roof["hbm"]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roof["hbm"]["frac"]})
        iso_stages = {}
        for st in ofk.STAGES:
            ms = iso[st][0] / args.steps
            tr = checked_traffic(st, B, ab[st] * B) if ms > 0 else None
            iso_stages[st] = {"ms_per_step": round(ms, 4),
                              "algorithmic_GBps": round(ab[st] * B / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                              "hbm_frac": round(ab[st] * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                              "traffic": tr}
            if tr is not None and tr < ab[st] * B:
                iso_stages[st]["note"] = "PMC traffic is below the algorithmic bytes: SURVEY's byte count is not this kernel's minimum (see north_star_kernel)"
        pyr_ms = iso["pyr"][0] / args.steps                      # every pyramid launch of a step (one at 1080p: levels 1-3 of both frames)
        pyr_tr = checked_traffic("pyr", B, ab["pyr"] * B)
        gb = lambda nbytes: round(nbytes / (pyr_ms * 1e-3) / 1e9, 1) if pyr_ms > 0 else None
        fr = lambda nbytes: round(nbytes / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if pyr_ms > 0 else None
        north_star_kernel = {"kernel": "k_pyr3_stream (pyramid levels 1-3 of both frames, one launch" + (")" if cfg.max_level <= 3 else "; deeper levels: k_pyr_down_stream)"),
                             "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_ms": round(pyr_ms, 4),
                             "achieved": gb(ab["pyr"] * B), "frac": fr(ab["pyr"] * B),
                             "algorithmic_bytes_per_launch": int(ab["pyr"] * B),
                             "achieved_fused_minimum": gb(ab["pyr_fused_minimum"] * B), "frac_fused_minimum": fr(ab["pyr_fused_minimum"] * B),
                             "fused_minimum_bytes_per_launch": int(ab["pyr_fused_minimum"] * B),
                             "traffic": pyr_tr, "achieved_traffic": gb(pyr_tr) if pyr_tr else None, "frac_traffic": fr(pyr_tr) if pyr_tr else None}
        t_grp = iso_ms["pyr"] * max(1, iso["pyr"][1] // max(1, args.steps)) + iso_ms["lk"]
        grp_bytes = (ab["pyr"] + ab["lk"]) * iso_pairs
        line = {
            "metric": f"frame-pairs/sec @{W}x{H} (LK+velocity)",
            "value": round(world * B * args.steps / dt, 2),
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/i32 image stages, f32 LK solve, f64 velocity solve",
            "data": "synthetic",
            "config": {"workload": C["workload"], "pairs_per_gpu_per_step": B, "streams_per_gpu": args.streams, "corners_mean": round(n_pts, 1), "candidates_mean": round(n_cand, 1), "win": cfg.win, "max_level": cfg.max_level,
                       "sharding": f"{world} x independent pair batches, RCCL all_gather of [B,8] f32 records on the library's stream" + (f", {comm.n_comms} communicator(s) per rank" if launched else " (single process: nothing to exchange)")},
            "roofline": roof,
            # "Pyramid+LK group (north-star kernel)" = G_pyr + G_lk (Scharr fused into LK), serial-pass durations
            "north_star_group": {"kernels": f"{KERNEL_OF['pyr']} + {KERNEL_OF['lk']}", "bound": "hbm",
                                 "achieved": round(grp_bytes / (t_grp * 1e-3) / 1e9, 1) if t_grp > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(grp_bytes / (t_grp * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t_grp > 0 else None,
                                 "algorithmic_bytes_per_launch": int(grp_bytes), "avg_ms": {"pyr": round(iso_ms["pyr"], 4), "lk": round(iso_ms["lk"], 4)},
                                 "traffic": (checked_traffic("pyr", iso_pairs, ab["pyr"] * iso_pairs) or 0) + (checked_traffic("lk", iso_pairs, ab["lk"] * iso_pairs) or 0) or None,
                                 "note": "LK is bound by VALU issue, not HBM (stages_isolated.lk, roofline of stage lk in DESIGN.md §4)"},
            "pipeline_algorithmic_GBps": round(sum(v for k, v in ab.items() if k != "pyr_fused_minimum") * world * B * args.steps / dt / 1e9, 1),
            # event brackets of every stage from K more steps of the SAME schedule run right after the timed region (the timed steps
            # carry no stage events): under overlap a bracket includes the other streams' kernels, it is not a kernel duration
            "stages": stages,
            # every stage alone on the chip (serial pass outside the timed region) and the HBM-bound group BASELINE.json's
            # target names (pyramid construction; LK itself is VALU-bound)
            "stages_isolated": iso_stages,
            # the HBM-bound member of that group alone.  Three fractions of the 8 TB/s peak, because SURVEY's 6.8 MB/pair (1080p) counts
            # re-reading levels 1 and 2, which a pass that builds all levels from registers never does: `frac` = SURVEY's formula,
            # `frac_fused_minimum` = by the bytes such a pass must move (level 0 in, levels 1..L out), `frac_traffic` = by the PMC bytes
            "north_star_kernel": north_star_kernel,
            "velocity_sample": [round(float(x), 6) for x in rec[0, :3]],
            "velocity_truth": list(truth["v"]),
        }
        if world == 1 and args.cpu_sample > 0:
            sample = min(args.cpu_sample, B)
            cps, v_cpu = cpu_baseline(prev, nxt, sensors, cfg, sample)
            # the same oracle on this GPU's share of the host (16 cores per GPU on the bench boxes), one pair per thread
            nthr = max(1, min(16, os.cpu_count() or 1))
            msample = min(B, 6 * nthr)
            cps_mt, _ = cpu_baseline(prev, nxt, sensors, cfg, msample, threads=nthr)
            line["cpu_baseline"] = {"value": round(cps, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
                                    "sample": f"{sample} of the same {W}x{H} pairs through oracle/ (C image stages + numpy lstsq), 1 thread, "
                                              f"host has {os.cpu_count()} cores",
                                    "velocity_max_rel_diff_vs_gpu": float(np.max(np.abs(v_cpu - rec[sample - 1, :3]) / np.abs(v_cpu))),
                                    "multi_thread": {"value": round(cps_mt, 3), "cores": nthr, "sample": f"{msample} pairs, one pair per thread"}}
        if world == 1 and not args.no_ingest:
            line["ingest_inclusive"] = ingest_inclusive(pipe, prev, nxt, sensors, B)
        if world == 1 and cname == "c4":
            line["monte_carlo_sweep"] = monte_carlo_sweep()
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()
    pipe.close()


if __name__ == "__main__":
    main()
