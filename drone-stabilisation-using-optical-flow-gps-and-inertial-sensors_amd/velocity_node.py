"""velocity_node.py — the reference's ROS node `velocity_measurment_node`, with its numpy loops and (commented-out)
OpenCV calls running on the HIP kernels.

Module level: generate_test_data (node:25-29), solve_lgs (node:30-42), feasible (node:44-50).
Class optical_fusion (node:52-267): same attributes, flags, callbacks (call_dist / call_imu / call_optical), topic
names and printed output.  Differences, all deliberate:
  * rospy is optional: `optical_fusion(spin=False)` builds the state without ROS so the node logic is testable;
    `step()` is one iteration of the reference's busy loop (node:226-267).
  * `synthetic_test=True` (default) reproduces the node AS SHIPPED: 4 hard-coded features (node:123), flow
    overwritten by generate_test_data (node:236), feasibility forced to -1 (node:240).  With
    `synthetic_test=False` the pipeline the reference keeps inside ''' blocks runs instead — on the DEVICE-RESIDENT
    stream loop (pipeline.FlowStream, ofk_stream_step_fused): the CompressedImage payload is decoded on the GPU, gray
    levels / pyramids / tracks never leave HBM, and one callback = JPEG decode, goodFeaturesToTrack on the first frame
    (node:120) or calcOpticalFlowPyrLK from the resident tracks (node:133-136), the real flow, the real r_tilde filter,
    solve_lgs, lever arm (node:229-258) and the re-detection with a circle mask (node:157-173); step() publishes it.
  * callbacks and step() snapshot shared state under one lock (the reference races, node:61-177 vs :226-267).
"""
import copy
import os
import threading
import time

import numpy as np

try:
    from . import ofk, cv2_hip as cv2, of_library as of
    from .pipeline import FlowStream, FusionConfig, PipelineConfig
except ImportError:
    import ofk
    import cv2_hip as cv2
    import of_library as of
    from pipeline import FlowStream, FusionConfig, PipelineConfig

try:                                    # ROS is optional (absent in this image)
    import rospy
    from cv_bridge import CvBridge
    from sensor_msgs.msg import CompressedImage, Image, Imu, Range
except ImportError:
    rospy = None
    CvBridge = CompressedImage = Image = Imu = Range = None


def generate_test_data(x, v, omega, d, n):
    """node:25-29 — forward flow model, one thread per point (k_flow_model)."""
    x = np.asarray(x, np.float64)
    if len(x) == 0:
        return np.zeros((0, 2))
    return ofk.default_context().flow_model(x[:, :2], v, omega, float(d), n)


def solve_lgs(x, u, d, n, omega):
    """node:30-42 — returns np.linalg.lstsq's 4-tuple (v, R, rank, s): R has shape (1,) when rank == 3 and
    3N > 3, else is empty.  On failure the reference returns the 2-tuple (zeros(3), 10000*ones(3N)) from a bare
    except (node:41-42); that path is kept for non-finite results."""
    x = np.asarray(x, np.float64); u = np.asarray(u, np.float64)
    out = ofk.default_context().velocity_solve(ofk.SOLVE_NODE, x[:, :2], u[:, :2], d=float(d), nrm=n, omega=omega)
    if not np.all(np.isfinite(out)):
        return np.zeros(3), 10000 * np.ones(3 * len(x))
    rank = int(out[4])
    R = np.array([out[3]]) if (rank == 3 and 3 * len(x) > 3) else np.empty(0)
    return out[:3].copy(), R, rank, out[5:8].copy()


def feasible(x, v, omega, T, u, d, n):
    """node:44-50 (reference not executable: uses v_cr/u_cr before definition, returns nothing).  The evident
    intent — parallelity and distance of each point given the lever-arm corrected velocity — is returned."""
    x = np.asarray(x, np.float64); u = np.asarray(u, np.float64)
    ve = np.asarray(v, np.float64) + np.cross(omega, T)
    v_cross = np.cross(np.concatenate([x[:, :2], np.ones((len(x), 1))], 1), ve[None, :])
    u_cross = np.cross(np.concatenate([x[:, :2], np.zeros((len(x), 1))], 1), np.concatenate([u[:, :2], np.zeros((len(x), 1))], 1))
    vn = np.linalg.norm(v_cross, axis=1); un = np.linalg.norm(u_cross, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        par = np.where(vn * un != 0, np.einsum("ni,ni->n", v_cross, u_cross) / (vn * un), 1.0)
        dist = np.where(un != 0, (np.concatenate([x[:, :2], np.ones((len(x), 1))], 1) @ np.asarray(n, np.float64)) * vn / un / d, 0.0)
    return par, dist


class optical_fusion:
    # parameters the reference sets inline (node:96-107, :182, :241)
    min_feat = 20
    max_feat = 100
    feature_params = dict(qualityLevel=0.7, minDistance=10, blockSize=12)
    lk_params = dict(winSize=(15, 15), maxLevel=3, criteria=(cv2.TERM_CRITERIA_EPS | cv2.TERM_CRITERIA_COUNT, 20, 0.03))
    scaling = 0.01
    T = 0.0


    # ---- the attributes call_imu owns (node:61-89).  Plain host values while the node runs as shipped; with the restored pipeline
    #      their truth lives on the device between frames and the host copy is refreshed only when somebody reads one
    #      (ofk_imu_state), or pushed down when somebody assigns one.
    def _imu_refresh(self):
        if self._imu_stale and self._stream is not None:
            self._imu_flush()
            st, _ = self._stream.ctx.imu_state(1)
            s0 = st[0]
            d = self._imu
            d["vel"] = s0[0:3].copy(); d["old_time"] = float(s0[3]); d["time_zero"] = int(s0[4]); d["first_imu_"] = bool(s0[5] != 0.0)
            d["rotation"] = s0[6:15].reshape(3, 3).copy(); d["normal"] = s0[15:18].copy(); d["ang"] = s0[18:21].copy(); d["ang_err"] = s0[21:24].copy()
            self._imu_stale = False

    def _imu_flush(self):
        """Host -> device: a state somebody assigned on the host first, then the messages received since the last frame."""
        fs = self._stream
        if fs is None:
            return
        if self._imu_host_dirty:
            d = self._imu
            s0 = np.zeros(ofk.IMU_STATE)
            s0[0:3] = d["vel"]; s0[3] = d["old_time"]; s0[4] = d["time_zero"]; s0[5] = 1.0 if d["first_imu_"] else 0.0
            s0[6:15] = np.asarray(d["rotation"], np.float64).ravel(); s0[15:18] = d["normal"]; s0[18:21] = d["ang"]
            s0[21:24] = np.asarray(d["ang_err"], np.float64).ravel()[:3] if np.size(d["ang_err"]) >= 3 else 0.0
            fs.ctx.imu_reset(1, s0)
            self._imu_host_dirty = False
        if self._imu_pending:
            fs.push_imu(np.stack(self._imu_pending)[None])
            self._imu_pending = []

    def _imu_get(self, name):
        with self._lock:
            self._imu_refresh()
            return self._imu[name]

    def _imu_set(self, name, value):
        with self._lock:
            self._imu_refresh()                                  # messages queued before the assignment are applied before it
            self._imu[name] = value
            if self._stream is not None:
                self._imu_host_dirty = True

    vel = property(lambda self: self._imu_get("vel"), lambda self, v: self._imu_set("vel", v))
    old_time = property(lambda self: self._imu_get("old_time"), lambda self, v: self._imu_set("old_time", v))
    time_zero = property(lambda self: self._imu_get("time_zero"), lambda self, v: self._imu_set("time_zero", v))
    first_imu_ = property(lambda self: self._imu_get("first_imu_"), lambda self, v: self._imu_set("first_imu_", v))
    rotation = property(lambda self: self._imu_get("rotation"), lambda self, v: self._imu_set("rotation", v))
    normal = property(lambda self: self._imu_get("normal"), lambda self, v: self._imu_set("normal", v))
    ang = property(lambda self: self._imu_get("ang"), lambda self, v: self._imu_set("ang", v))
    ang_err = property(lambda self: self._imu_get("ang_err"), lambda self, v: self._imu_set("ang_err", v))

    def call_dist(self, data):
        """node:55-58 — reads the range and discards it (self.d stays 0.75)."""
        distance = data.range  # noqa: F841

    def call_imu(self, data):
        """node:61-89 — quaternion -> R, normal, dead-reckoning between optical fixes (k_imu)."""
        w, q, a, st = data.angular_velocity, data.orientation, data.linear_acceleration, data.header.stamp
        cov = data.angular_velocity_covariance
        msg = np.array([st.secs, st.nsecs, q.x, q.y, q.z, q.w, w.x, w.y, w.z, cov[0], cov[4], cov[8], a.x, a.y, a.z], np.float64)
        with self._lock:
            if self._stream is not None and not self.synthetic_test:
                # The restored pipeline keeps node:61-89's state on the device (FlowStream, FusionConfig.node(): k_imu_seq applies the
                # messages, k_stream_fuse reads normal / omega / R / velocity from there and writes self.vel = v_uav back).  A message
                # costs a list append here; the whole batch since the last frame goes up with that frame (or when an attribute is read).
                self._imu_pending.append(msg)
                self._imu_stale = True
                self.got_ang_vel_ = True
                return
            state = np.zeros(ofk.IMU_STATE)
            state[0:3] = self.vel; state[3] = self.old_time; state[4] = self.time_zero; state[5] = 1.0 if self.first_imu_ else 0.0
            state = self._ctx().imu_propagate(state, msg)
            if not self.got_vel_:                       # node:80 (got_vel_ is never set on self, node:260-262 toggles a local)
                self.vel = state[0:3].copy()
            self.old_time = float(state[3]); self.time_zero = int(state[4]); self.first_imu_ = False
            self.rotation = state[6:15].reshape(3, 3).copy(); self.normal = state[15:18].copy()
            self.ang = state[18:21].copy(); self.ang_err = state[21:24].copy()
            self.got_ang_vel_ = True

    def _decode(self, image_raw):
        """node:112 (cv_bridge.compressed_imgmsg_to_cv2(image_raw, 'bgr8')).  A sensor_msgs/CompressedImage - anything with a
        `.data` byte payload - or a bare JPEG byte string is decoded on the GPU (cv2_hip.imdecode = what cv_bridge calls);
        cv_bridge only takes over for payloads the device decoder does not handle (png, progressive JPEG)."""
        if isinstance(image_raw, np.ndarray):
            return image_raw
        data = image_raw if isinstance(image_raw, (bytes, bytearray, memoryview)) else getattr(image_raw, "data", None)
        if data is not None:
            img = cv2.imdecode(np.frombuffer(bytes(data), np.uint8), cv2.IMREAD_COLOR)
            if img is not None:
                return img
        if CvBridge is None:
            raise RuntimeError("call_optical: not a baseline JPEG payload and cv_bridge is not available - pass BGR numpy frames")
        return CvBridge().compressed_imgmsg_to_cv2(image_raw, 'bgr8')

    def call_optical(self, image_raw):
        """node:92-177 - image callback.  As shipped (synthetic_test) it only decodes, converts and stores the frame; with the
        reference's commented-out pipeline restored the whole frame step runs on the device (_call_optical_resident)."""
        with self._lock:
            if self.got_picture_:
                return
            if not self.synthetic_test:
                return self._call_optical_resident(image_raw)
            image = self._decode(image_raw)
            image_gray = cv2.cvtColor(image, cv2.COLOR_BGR2GRAY)
            if self.first:
                first_feat = np.array([[-401, 300], [399, -300], [400, 301], [-400, -299]])           # node:123
                self.feat = np.asarray(first_feat).reshape((len(first_feat), 2))
                self.feat_err = np.zeros(len(first_feat))
            else:
                self.init = False
                self.got_picture_ = True
            self.old_pic = image_gray
            self.first = False

    def _payload(self, image_raw):
        """The frame as the device wants it: a baseline-JPEG byte string (decoded on the GPU, nothing decoded crosses PCIe) or an
        HxWx3 BGR array.  Anything the device decoder refuses goes through cv_bridge like in the reference (node:112)."""
        if isinstance(image_raw, np.ndarray):
            return image_raw
        data = image_raw if isinstance(image_raw, (bytes, bytearray, memoryview)) else getattr(image_raw, "data", None)
        if data is not None:
            data = bytes(data)
            try:
                h, w, _ = ofk.jpeg_info(data)
                if h <= ofk.MAX_DIM and w <= ofk.MAX_DIM:
                    return data
            except ofk.OfkError:
                pass
        return self._decode(image_raw)

    def _call_optical_resident(self, image_raw):
        """node:112-175 with the commented-out blocks restored, on pipeline.FlowStream: the frame (JPEG or BGR) goes to the device
        once; gray levels, pyramids and the tracks stay in HBM between callbacks.  First frame: goodFeaturesToTrack (:120).  Later
        frames: calcOpticalFlowPyrLK from the resident tracks (:133-136), centre + scale, r_tilde filter with the dead-reckoned
        velocity, solve_lgs, lever arm + rotation (:229-258, the main loop's work, one kernel behind LK), status filter,
        re-detection with a disc mask when <= min_feat tracks are left (:157-166), frame swap (:175).  step() then publishes what
        was computed."""
        frame = self._payload(image_raw)
        is_jpeg = not isinstance(frame, np.ndarray)
        h, w = ofk.jpeg_info(frame)[:2] if is_jpeg else frame.shape[:2]
        if self._stream is None or self._stream_dim != (h, w):
            if self._stream is not None:
                self._stream.close()
            cnt, eps = cv2._criteria(self.lk_params["criteria"])
            cfg = PipelineConfig(max_corners=int(self.max_feat), quality=float(self.feature_params["qualityLevel"]),
                                 min_distance=float(self.feature_params["minDistance"]), block_size=int(self.feature_params["blockSize"]),
                                 win=int(self.lk_params["winSize"][0]), max_level=int(self.lk_params["maxLevel"]), max_count=cnt, eps=eps,
                                 use_feasibility=True, feas_T=float(self.T))
            self._stream = FlowStream(w, h, batch=1, cfg=cfg, device=int(os.environ.get("OFK_DEVICE", "0")), min_features=int(self.min_feat),
                                      mask_radius=30, fusion=FusionConfig.node())
            self._stream_dim = (h, w)
            self.first = True
            self._imu_host_dirty = True                          # the dead-reckoning state so far moves to the device with the first step
        fs = self._stream
        if self.first:
            tracks, counts = fs.begin_jpeg([frame]) if is_jpeg else fs.begin(frame[None])
            n = int(counts[0])
            self.feat = tracks[0, :n].astype(np.float64).reshape(n, 2)
            self.feat_err = np.zeros(n)
            self.first = False
            return
        old = np.asarray(self.feat, np.float32).reshape(-1, 2)
        translation = of.pix_trans((320, 240))                   # node:229 (the node centres with (160, 120) whatever the frame size)
        fs._params.feas_T = float(self.T)
        self._imu_flush()                                        # host-side assignments, then the IMU messages since the last frame: one upload
        # normal / omega / R / prior velocity come from the resident IMU state (FusionConfig.node(): use_imu); the record only
        # carries what call_imu does not own
        sensors = ofk.make_sensors(1, d=self.d, offset=self.offset, scaling=self.scaling, cx=translation[0], cy=translation[1])
        rec, fused, tracks, counts = fs.step_fused([frame] if is_jpeg else frame[None], sensors)
        self._imu_stale = True                                   # the step wrote self.vel = v_uav on the device (node:261)
        nxt, keep = fs.ctx.stream_last_points(max(1, len(old)))
        k = keep[0, :len(old)] != 0
        n = int(counts[0])
        self.flow = (nxt[0, :len(old)][k] - old[k]).reshape(-1, 1, 2)                               # node:136
        self.feat = tracks[0, :n].astype(np.float64).reshape(n, 2)                                   # kept points (+ re-detected ones, node:166)
        self.feat_err = np.zeros(n)
        r = rec[0]
        self._resident_result = (r[0:3].copy(), r[8:11].copy(), (np.array([r[3]]) if (r[4] == 3 and 3 * r[11] > 3) else np.empty(0)), int(r[4]),
                                 r[5:8].copy()) if r[15] else None
        self.init = False
        self.got_picture_ = True

    def _ctx(self):
        return ofk.default_context()

    def step(self):
        """One pass of the reference's main loop body (node:226-267).  Returns v_obs or None."""
        with self._lock:
            if not (self.got_picture_ and not self.init):
                return None
            if not self.synthetic_test:                          # the callback already ran the loop body on the device (node:229-258)
                res, self._resident_result = self._resident_result, None
                self.got_picture_ = False
                if res is None:
                    return None
                v_obs, v_uav, R, rank, sv = res
                print('    '.join(map(str, v_obs)))                                                          # node:259
                # node:261 self.vel = v_uav happened on the device behind the solve (vel_overwrite); reading self.vel fetches it
                self.last_v_uav = v_uav
                self.last_residual, self.last_rank, self.last_s = R, rank, sv
                return v_obs
            # ---- the node as shipped: 4 test features, flow from generate_test_data, feasibility forced to -1
            translation = of.pix_trans((320, 240))
            x = copy.deepcopy(self.feat).astype(float)
            x[:, 0] = (x[:, 0] - translation[0]) * self.scaling
            x[:, 1] = (x[:, 1] - translation[1]) * self.scaling
            u = generate_test_data(x, np.array([1, 1, 1]), np.array([0, 0, 0]), self.d, np.array([0, 0, 1]))       # node:236
            feasibility, dummy_d = of.r_tilde(x, u, self.normal, self.vel, self.d)                                  # node:238
            feasibility = -1 * np.ones(len(x))                                                                      # node:240
            keep = feasibility <= self.T
            x = x[keep]; u = u[keep]; dummy_d = dummy_d[keep]
            self.feat = self.feat[keep]
            v_obs = None
            if len(x) >= 3:
                v_obs, R, rank, s = solve_lgs(x, u, self.d, self.normal, self.ang)
                v_uav = self._ctx().post_solve(v_obs, self.rotation, self.ang, self.offset)                        # node:258
                print('    '.join(map(str, v_obs)))
                self.vel = v_uav
                self.last_residual, self.last_rank, self.last_s = R, rank, s
            self.got_picture_ = False
            return v_obs

    def __init__(self, spin=True, synthetic_test=True):
        self._lock = threading.RLock()
        self._imu = {}                                           # host copy of the attributes call_imu owns (see the properties above)
        self._imu_pending, self._imu_stale, self._imu_host_dirty = [], False, False
        self._stream = None
        self.synthetic_test = synthetic_test
        self._stream_dim = None                                  # (self._stream: pipeline.FlowStream of the restored pipeline, created with the first frame)
        self._resident_result = None
        self.vel = np.array([0.1, 0.1, 0.1])
        self.vel_err = np.array([0.1, 0.1, 0.1])
        self.feat = np.ones((1, 2))
        self.feat_err = np.ones((1, 1))
        self.flow = np.zeros((1, 1, 2))
        self.flow_err = np.zeros((1, 1, 2))
        self.ang = np.array([0, 0, 0])
        self.ang_err = np.zeros((3, 3))
        self.d = 0.75
        self.d_err = 0
        self.rotation = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]])
        self.old_pic = np.zeros((480, 640))
        self.old_time = 0
        self.time_zero = 0
        self.offset = np.array([0, 0, 0.1])
        self.normal = np.array([0, 0, 1])
        self.normal_err = np.array([0, 0, 1])
        self.init = True
        self.first = True
        self.first_imu_ = True
        self.got_normal_ = False
        self.got_vel_ = False
        self.got_picture_ = False
        self.got_ang_vel_ = False
        if not spin:
            return
        if rospy is None:
            raise RuntimeError("rospy is not installed: construct optical_fusion(spin=False) and drive call_* / step() yourself")
        rospy.Subscriber('/mavros/imu/data', Imu, self.call_imu)
        rospy.Subscriber('/camerav2_1280x960/image_raw/compressed', CompressedImage, self.call_optical)
        rospy.Subscriber('/mavros/distance_sensor/hrlv_ez4_pub', Range, self.call_dist)
        self.visualize = rospy.Publisher('visualisation', Image, queue_size=1)
        while not rospy.is_shutdown():
            if self.step() is None:
                time.sleep(0.0005)          # the reference spins hot; yield the GIL to the callback threads


def main():
    print('++')
    if rospy is None:
        raise SystemExit("velocity_measurment_node needs ROS (rospy, cv_bridge, sensor_msgs)")
    rospy.init_node('velocity_calc')
    print('--')
    try:
        optical_fusion()
    except rospy.ROSInterruptException:
        pass


if __name__ == '__main__':
    main()
