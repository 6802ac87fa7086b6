"""pipeline.py — batched, HBM-resident frame-pair pipeline: BGR pair -> gray -> pyramids -> Shi-Tomasi corners
-> pyramidal LK -> centre/scale -> (feasibility) -> least-squares body velocity -> lever arm + rotation.

It strings together, for a batch of independent frame pairs, exactly the stage order of the reference's
offline prototype (optical_flow_experiments/of_module.py:36-152) and of the node's main loop
(velocity_measurment_node:226-261), every stage a HIP kernel (libofk.so, ofk_pairs_run).
"""
from dataclasses import dataclass

import numpy as np

try:
    from . import ofk
except ImportError:      # package directory put on sys.path directly
    import ofk


@dataclass
class PipelineConfig:
    max_corners: int = 500
    quality: float = 0.01
    min_distance: float = 10.0
    block_size: int = 7
    win: int = 15
    max_level: int = 3
    max_count: int = 20
    eps: float = 0.03
    min_eig_thr: float = 1e-4
    solve_variant: int = ofk.SOLVE_NODE
    use_feasibility: bool = False
    feas_T: float = 0.0

    # the three parameter sets the reference carries inline
    @classmethod
    def node(cls):               # velocity_measurment_node:96-107
        return cls(max_corners=100, quality=0.7, min_distance=10, block_size=12, win=15, max_level=3, max_count=20, eps=0.03)

    @classmethod
    def of_module(cls):          # optical_flow_experiments/of_module.py:12-23, T = 0.9 (:55), the inline [p]x / dist_i system (:136-146)
        return cls(max_corners=50, quality=0.3, min_distance=20, block_size=32, win=15, max_level=3, max_count=10, eps=0.5,
                   solve_variant=ofk.SOLVE_OFMODULE, feas_T=0.9)

    @classmethod
    def evaluate_exp(cls):       # flight_experiments/evaluate_exp.py:37-48
        return cls(max_corners=20, quality=0.7, min_distance=10, block_size=7, win=15, max_level=3, max_count=20, eps=0.03)

    @classmethod
    def baseline_1080p(cls):     # BASELINE.json configs[1]: 500 corners, 3-level pyramid
        return cls(max_corners=500, quality=0.01, min_distance=10, block_size=7, win=15, max_level=3, max_count=20, eps=0.03)

    def to_params(self):
        return ofk.Params(int(self.max_corners), float(self.quality), float(self.min_distance), int(self.block_size),
                          int(self.win), int(self.max_level), int(self.max_count), float(self.eps), float(self.min_eig_thr),
                          int(self.solve_variant), 1 if self.use_feasibility else 0, float(self.feas_T))


@dataclass
class FilterModel:
    """Matrices of the per-stream linear Kalman filter (k_kf; ofk_kf_predict_update / the resident filter of FlowStream).

    kf3()  the reference's filter, of_module.py:63-76: cv2.KalmanFilter(3, 3, 0) with F = B = H = I, Q = 1e-5 I, R = 10 I,
           P0 = 0.1 I, x0 = 0; predict(control) every frame (:122), correct(-v_obs) (:152).
    ekf6() the build-defined 6-state superset BASELINE.json configs[2] asks for (the reference has no such filter and no GPS
           data: DESIGN.md §2a): x = [v, b] with b the accelerometer bias in the velocity frame,
               v' = v + u - b dt,   b' = b          F = [[I, -dt I], [0, I]],  B = [[I], [0]]
           where the control u is the dead-reckoning increment R (a - 9.81 n) dt of node:80-82; the optical fix measures v
           (H = [I 0], z = -v_obs as in the reference); with gps=True a second velocity measurement (what a GPS receiver's
           velocity output would deliver) is stacked under it: H = [[I 0], [I 0]], R = diag(r I, r_gps I).  Both models are
           linear, so the "EKF" is the Kalman filter itself; kf3 is its restriction to the first three states, and with
           dt = 0 and no bias noise the v-block of ekf6 reproduces kf3 bit for bit (tested).
    """
    F: np.ndarray
    B: np.ndarray
    H: np.ndarray
    Q: np.ndarray
    R: np.ndarray
    P0: np.ndarray
    x0: np.ndarray

    @property
    def ns(self):
        return self.F.shape[0]

    @property
    def nm(self):
        return self.H.shape[0]

    @property
    def nc(self):
        return self.B.shape[1]

    @classmethod
    def kf3(cls):
        I = np.eye(3)
        return cls(F=I.copy(), B=I.copy(), H=I.copy(), Q=1e-5 * I, R=10.0 * I, P0=0.1 * I, x0=np.zeros(3))

    @classmethod
    def ekf6(cls, dt=1.0 / 30.0, q_v=1e-5, q_b=1e-7, r=10.0, p0=0.1, gps=False, r_gps=1.0):
        I, Z = np.eye(3), np.zeros((3, 3))
        F = np.block([[I, -dt * I], [Z, I]])
        Bm = np.vstack([I, Z])
        H = np.hstack([I, Z])
        R = r * I
        if gps:
            H = np.vstack([H, np.hstack([I, Z])])
            R = np.block([[r * I, Z], [Z, r_gps * I]])
        return cls(F=F, B=Bm, H=H, Q=np.diag([q_v] * 3 + [q_b] * 3), R=R, P0=p0 * np.eye(6), x0=np.zeros(6))


@dataclass
class FusionConfig:
    """What FlowStream.step_fused does between LK and the next frame (ofk_fusion, include/ofk.h)."""
    use_imu: bool = False
    flow: int = ofk.FLOW_LK
    keep: int = ofk.KEEP_STATUS
    filter: bool = False
    control: int = ofk.CONTROL_SENSORS
    z_sign: float = 1.0
    z_source: int = 0
    vel_overwrite: bool = False
    redetect_replace: bool = False
    min_solve: int = 2
    hold_on_skip: bool = False
    model: "FilterModel" = None

    @classmethod
    def of_module(cls, synthetic_flow=True, hold_on_skip=False):
        """The loop of optical_flow_experiments/of_module.py:78-167: kalman.predict(control) -> legacy r_tilde with the predicted
        velocity, keep r - (status - 1) >= T with cv2's uint8 status (a lost point's status-1 wraps to 255: dropped) -> A_i = [p]x / dist_i system -> kalman.correct(-v_obs); tracks := kept points;
        <= 10 tracks: replace by fresh corners.  synthetic_flow reproduces :113-114 (the measured flow overwritten by the
        rotational field of a random omega); False keeps the LK flow the script's TODO asks for.
        Use with PipelineConfig.of_module() (feas_T = 0.9, solve_variant = OFMODULE) and sensors cx, cy = of.pix_trans((480, 640)),
        scaling = 1 (the script works in pixels, :96-102).  hold_on_skip=True (a FlowStream of ONE stream, which is what the script is)
        also reproduces its `continue` at :138: a frame with <= 3 feasible points leaves old_gray and old_pos untouched, so the next
        frame is tracked from the old one; in a batch of streams the frame always advances (one frame swap for all)."""
        return cls(flow=ofk.FLOW_ROTATIONAL if synthetic_flow else ofk.FLOW_LK, keep=ofk.KEEP_LEGACY, filter=True, control=ofk.CONTROL_SENSORS,
                   z_sign=-1.0, z_source=0, redetect_replace=True, min_solve=3, hold_on_skip=bool(hold_on_skip), model=FilterModel.kf3())

    @classmethod
    def node(cls):
        """velocity_measurment_node: IMU dead-reckoning between optical fixes (node:61-89), r_tilde filter with the dead-reckoned
        velocity (:238-245), solve (:257), self.vel = v_uav (:261).  No Kalman filter (the reference node has none)."""
        return cls(use_imu=True, keep=ofk.KEEP_STATUS, filter=False, vel_overwrite=True, min_solve=2)

    @classmethod
    def ekf6(cls, dt=1.0 / 30.0, gps=False, **kw):
        """BASELINE configs[2]/[3]'s "6-state EKF": the node's loop with the build-defined filter FilterModel.ekf6 in place of the
        hard overwrite: predict with the IMU's velocity increments since the last frame, correct with +v_uav (world frame)."""
        return cls(use_imu=True, keep=ofk.KEEP_STATUS, filter=True, control=ofk.CONTROL_IMU, z_sign=1.0, z_source=1, vel_overwrite=False,
                   min_solve=2, model=FilterModel.ekf6(dt=dt, gps=gps, **kw))

    def to_struct(self):
        return ofk.Fusion(1 if self.use_imu else 0, int(self.flow), int(self.keep), 1 if self.filter else 0, int(self.control), float(self.z_sign),
                          int(self.z_source), 1 if self.vel_overwrite else 0, 1 if self.redetect_replace else 0, int(self.min_solve), 1 if self.hold_on_skip else 0)


class FlowStream:
    """`batch` independent video streams with persistent tracks on the device: the loop of velocity_measurment_node:92-177
    (commented-out blocks restored) / of_module.py:78-167 / evaluate_exp.py:77-121, one frame per `step`.

        fs = FlowStream(1280, 960, batch=1, cfg=PipelineConfig.node(), min_features=20, mask_radius=30)
        tracks, counts = fs.begin(first_frames)                 # goodFeaturesToTrack on the first frames
        records, tracks, counts = fs.step(frames, sensors)      # LK, velocity, status filter, re-detection, frame swap
    """

    def __init__(self, width, height, batch=1, cfg=None, device=0, min_features=20, mask_radius=30, fusion=None):
        self.cfg = cfg or PipelineConfig.node()
        self.batch = batch
        self.min_features, self.mask_radius = int(min_features), int(mask_radius)
        self.ctx = ofk.Context(device, width, height, batch, max(1, self.cfg.max_corners), max(0, self.cfg.max_level))
        self._params = self.cfg.to_params()
        self.fusion = fusion
        if fusion is not None:                                  # the per-stream filter state lives on the device from here on
            self._fusion = fusion.to_struct()
            self.ctx.imu_reset(batch)
            if fusion.filter:
                self.ctx.filter_configure(fusion.model or FilterModel.kf3(), batch)

    def push_imu(self, msgs, counts=None):
        """The IMU messages received since the last frame ([B, M, 15], layout ofk.h OFK_IMU_MSG), applied to the resident state."""
        self.ctx.imu_push(msgs, counts)

    def step_fused(self, next_frames, sensors):
        """One frame with the filters in the loop (ofk_stream_step_fused): BGR frames [B,h,w,3] or a list of B JPEG streams.
        Returns (records, fused [B,8] = filter state x[0..5], trace P, solved, tracks, counts)."""
        if self.fusion is None:
            raise ValueError("construct FlowStream(..., fusion=FusionConfig...) for step_fused")
        return self.ctx.stream_step_fused(next_frames, sensors, self._params, self._fusion, self.min_features, self.mask_radius)

    def begin(self, first_bgr):
        return self.ctx.stream_begin(first_bgr, self._params)

    def step(self, next_bgr, sensors):
        return self.ctx.stream_step(next_bgr, sensors, self._params, self.min_features, self.mask_radius)

    def begin_jpeg(self, first_streams):
        """begin() with the frames as the node receives them: one JPEG stream (CompressedImage payload) per camera."""
        return self.ctx.stream_begin_jpeg(first_streams, self._params)

    def step_jpeg(self, next_streams, sensors):
        return self.ctx.stream_step_jpeg(next_streams, sensors, self._params, self.min_features, self.mask_radius)

    def close(self):
        self.ctx.close()


def auto_streams(width, height, batch):
    """Free-running slices of ofk_pairs_run for a frame size (streams=None): ONE stage chain for large frames - the response kernel
    is the longest stage and wave-sized selection / solve workgroups fit beside it - and TWO for frames up to 640 x 480 in batches of
    at least 256 pairs, where LK is the longest stage, the latency-bound phases (selection, solve) weigh more and a second chain fills
    them: 1024 pairs of 640 x 480: 343-359 k pairs/s on one slice, 380-400 k on two; 1080p and 4K: one slice is 3-6 % faster."""
    return 2 if width * height <= 640 * 480 and batch >= 256 else 1


class FlowPipeline:
    def __init__(self, width, height, batch=1, cfg=None, device=0, streams=1):
        self.cfg = cfg or PipelineConfig.baseline_1080p()
        self.batch = batch
        if streams is None:
            streams = auto_streams(width, height, batch)
        self.streams = streams
        self.ctx = ofk.Context(device, width, height, batch, max(1, self.cfg.max_corners), max(0, self.cfg.max_level))
        self._params = self.cfg.to_params()
        if streams > 1:
            self.ctx.set_streams(streams)

    def upload(self, prev_bgr, next_bgr, sensors):
        self.ctx.pairs_upload(prev_bgr, next_bgr)
        self.ctx.pairs_set_sensors(sensors)

    def upload_jpeg(self, prev_streams, next_streams, sensors):
        """The same from baseline JPEG streams (CompressedImage payloads): decoded on the device, 15-20x less PCIe traffic."""
        self.ctx.pairs_upload_jpeg(prev_streams, next_streams)
        self.ctx.pairs_set_sensors(sensors)

    def run_jpeg_batches(self, batches, sensors, on_step=None):
        """The pipeline over an iterable of (prev_streams, next_streams) batches with the ingest double-buffered: a helper thread
        parses, stages and uploads batch k + 1 (ofk_jpeg_stage, slot (k + 1) & 1) while this thread decodes batch k on the device
        (ofk_pairs_upload_staged) and queues its ofk_pairs_run.  `sensors`: one [B, 28] array for all batches or a callable
        k -> array.  on_step(k) is called after step k was queued (e.g. to pick up step k - 1's records).  Returns the number of steps;
        the last one is still running (call sync() / run-download)."""
        from concurrent.futures import ThreadPoolExecutor
        it = iter(batches)
        first = next(it, None)
        if first is None:
            return 0
        n = 0
        with ThreadPoolExecutor(1) as ex:
            fut = ex.submit(self.ctx.jpeg_stage, 0, list(first[0]) + list(first[1]))
            while fut is not None:
                staged = fut.result()
                nxt = next(it, None)
                fut = ex.submit(self.ctx.jpeg_stage, (n + 1) & 1, list(nxt[0]) + list(nxt[1])) if nxt is not None else None
                self.ctx.pairs_upload_staged(n & 1, staged)
                if callable(sensors) or n == 0:
                    self.ctx.pairs_set_sensors(sensors(n) if callable(sensors) else sensors)
                self.run_async()
                if on_step is not None:
                    on_step(n)
                n += 1
        return n

    def run_async(self):
        self.ctx.pairs_run(self._params)

    def sync(self):
        self.ctx.sync()

    def run(self, points=True):
        self.run_async()
        return self.ctx.pairs_download(points=points)

    def close(self):
        self.ctx.close()
