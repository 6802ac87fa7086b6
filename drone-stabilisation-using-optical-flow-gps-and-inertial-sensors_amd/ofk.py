"""ofk.py — ctypes binding of libofk.so (include/ofk.h), numpy in / numpy out.

This is the only place the package touches native code.  There is no CPU fallback: if the
shared library is missing or no gfx950 device is usable, importing succeeds (so that
CPU-only tooling can inspect the package) but the first call raises OfkError.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libofk.so")

OK, E_INVALID, E_HIP, E_CAPACITY, E_NOGPU = 0, -1, -2, -3, -4
SOLVE_NODE, SOLVE_SIM, SOLVE_OFMODULE = 0, 1, 2
FEAS_RTILDE, FEAS_LEGACY, FEAS_SIM = 0, 1, 2
SENSOR_DOUBLES, RECORD_DOUBLES, SOLVE_DOUBLES = 28, 16, 8
IMU_STATE, IMU_MSG = 24, 15
MAX_DIM = 16384          # ofk_create rejects larger frames
STAGES = ("gray", "pyr", "eig", "nms", "select", "lk", "solve")

# every entry point include/ofk.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = (
    "ofk_version", "ofk_last_error", "ofk_device_count", "ofk_create", "ofk_destroy", "ofk_sync", "ofk_device_sync",
    "ofk_gray_bgr8", "ofk_pyr_down_u8", "ofk_pyramid_u8", "ofk_scharr_s16", "ofk_mineig_response", "ofk_select_corners",
    "ofk_good_features", "ofk_lk_pyr", "ofk_flow_model", "ofk_feasibility", "ofk_velocity_solve", "ofk_imu_propagate",
    "ofk_post_solve", "ofk_kf_predict_update", "ofk_of_simulation", "ofk_of_simulation_rng", "ofk_noise_normals", "ofk_feas_simulation", "ofk_hist_overlap", "ofk_associate_sensors", "ofk_feature_eval", "ofk_d_split", "ofk_pairs_upload", "ofk_pairs_upload_jpeg", "ofk_jpeg_stage", "ofk_jpeg_stage_error", "ofk_pairs_upload_staged", "ofk_jpeg_info", "ofk_jpeg_destuff", "ofk_jpeg_decode_bgr8", "ofk_pairs_set_sensors",
    "ofk_pairs_run", "ofk_pairs_download", "ofk_pairs_export_records_f32", "ofk_stream_begin", "ofk_stream_step",
    "ofk_stream_begin_jpeg", "ofk_stream_step_jpeg",
    "ofk_set_streams", "ofk_set_overlap", "ofk_set_tuning", "ofk_get_tuning", "ofk_mark", "ofk_mark_wait", "ofk_profile_enable", "ofk_profile_read", "ofk_resident_pyramid",
    "ofk_imu_reset", "ofk_imu_push", "ofk_imu_state", "ofk_filter_configure", "ofk_filter_state", "ofk_stream_step_fused",
    "ofk_stream_step_fused_jpeg", "ofk_stream_last_points", "ofk_pairs_filter_step",
    "ofk_comm_unique_id", "ofk_comm_init", "ofk_comm_add", "ofk_comm_destroy", "ofk_comm_rank", "ofk_comm_world", "ofk_comm_gather_records",
    "ofk_comm_fetch_records", "ofk_comm_allreduce_f64", "ofk_comm_count", "ofk_comm_pending", "ofk_comm_reorder_records",
)


class OfkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libofk error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("max_corners", C.c_int), ("quality", C.c_double), ("min_distance", C.c_double), ("block_size", C.c_int),
                ("win", C.c_int), ("max_level", C.c_int), ("max_count", C.c_int), ("eps", C.c_double),
                ("min_eig_thr", C.c_double), ("solve_variant", C.c_int), ("use_feasibility", C.c_int), ("feas_T", C.c_double)]


FLOW_LK, FLOW_ROTATIONAL = 0, 1
KEEP_STATUS, KEEP_LEGACY = 0, 1
CONTROL_SENSORS, CONTROL_IMU = 0, 1


class Fusion(C.Structure):
    """ofk_fusion (include/ofk.h): what ofk_stream_step_fused does between LK and the next frame."""
    _fields_ = [("use_imu", C.c_int), ("flow", C.c_int), ("keep", C.c_int), ("filter", C.c_int), ("control", C.c_int),
                ("z_sign", C.c_double), ("z_source", C.c_int), ("vel_overwrite", C.c_int), ("redetect_replace", C.c_int),
                ("min_solve", C.c_int), ("hold_on_skip", C.c_int)]


_lib = None
_lib_lock = threading.Lock()


def jpeg_info(stream):
    """(h, w, components) of a JPEG stream the device decoder accepts; OfkError otherwise (host-side header parse)."""
    data = bytes(stream)
    h, w, n = C.c_int(), C.c_int(), C.c_int()
    rc = load_library().ofk_jpeg_info(data, len(data), C.byref(h), C.byref(w), C.byref(n))
    if rc != 0:
        raise OfkError(rc, "not a JPEG stream the decoder supports (8-bit baseline Huffman, one interleaved scan, gray or YCbCr "
                           "4:4:4/4:2:2/4:2:0)")
    return h.value, w.value, n.value


def jpeg_destuff(stream):
    """(entropy segment without its byte stuffing and restart markers, offsets behind the removed RSTn markers) - what the ingest's
    staging hands to the device decoders (host only)."""
    data = bytes(stream)
    out = C.create_string_buffer(max(1, len(data)))
    rst = (C.c_uint32 * (len(data) // 2 + 1))()
    n, nr = C.c_size_t(), C.c_int()
    rc = load_library().ofk_jpeg_destuff(data, len(data), out, len(data), C.byref(n), rst, len(rst), C.byref(nr))
    if rc != 0:
        raise OfkError(rc, "ofk_jpeg_destuff: not a JPEG stream the decoder supports")
    return out.raw[:n.value], list(rst[:nr.value])


def load_library():
    """Loads libofk.so and declares the signatures.  Raises OfkError if the library is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise OfkError(E_NOGPU, f"{LIB_PATH} not built (run `make -C {os.path.join(_HERE, 'csrc')}` or "
                                    "__graft_entry__.build()); there is no CPU fallback")
        # The resident pipeline keeps 4 HIP streams busy (2 free-running slices x {chain, auxiliary}) beside the caller's own
        # (torch's, RCCL's).  The HIP runtime multiplexes streams onto 4 hardware queues unless told otherwise, and two streams of
        # one queue run in order.  Only effective if the runtime is not initialised yet; a value the user set wins.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        L = C.CDLL(LIB_PATH)
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.ofk_version.restype = i
        L.ofk_last_error.restype = C.c_char_p; L.ofk_last_error.argtypes = [vp]
        L.ofk_device_count.restype = i
        L.ofk_create.argtypes = [i, i, i, i, i, i, C.POINTER(vp)]
        L.ofk_destroy.argtypes = [vp]; L.ofk_sync.argtypes = [vp]
        L.ofk_gray_bgr8.argtypes = [vp, vp, i, i, i, vp]
        L.ofk_pyr_down_u8.argtypes = [vp, vp, i, i, i, vp]
        L.ofk_pyramid_u8.argtypes = [vp, vp, i, i, i, i, vp, C.POINTER(i)]
        L.ofk_scharr_s16.argtypes = [vp, vp, i, i, i, vp]
        L.ofk_mineig_response.argtypes = [vp, vp, i, i, i, i, vp]
        L.ofk_select_corners.argtypes = [vp, vp, vp, i, i, i, i, d, d, vp, vp]
        L.ofk_good_features.argtypes = [vp, vp, vp, i, i, i, i, d, d, i, vp, vp]
        L.ofk_lk_pyr.argtypes = [vp, vp, vp, i, i, i, vp, vp, i, i, i, i, d, d, vp, vp, vp]
        L.ofk_flow_model.argtypes = [vp, vp, i, i, vp, vp, vp, vp, vp, vp]
        L.ofk_feasibility.argtypes = [vp, i, vp, vp, i, i, vp, vp, vp, vp, vp, vp, vp]
        L.ofk_velocity_solve.argtypes = [vp, i, vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp]
        L.ofk_imu_propagate.argtypes = [vp, vp, vp, i]
        L.ofk_post_solve.argtypes = [vp, vp, vp, vp, vp, i, vp]
        L.ofk_associate_sensors.argtypes = [vp, vp, i, vp, vp, vp, i, vp, vp, i, vp, vp, vp]
        L.ofk_feature_eval.argtypes = [vp, vp, vp, vp, vp, vp, i, i, vp, vp, d, d, i, i, vp, vp, vp, vp, vp, vp, vp]
        L.ofk_d_split.argtypes = [vp, vp, vp, i, i, d, vp, vp, vp]
        L.ofk_kf_predict_update.argtypes = [vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i]
        L.ofk_of_simulation.argtypes = [vp, vp, vp, vp, vp, i, vp, i, vp, vp]
        L.ofk_of_simulation_rng.argtypes = [vp, vp, vp, vp, vp, i, C.c_ulonglong, C.c_uint, C.c_uint, i, vp, vp]
        L.ofk_noise_normals.argtypes = [vp, C.c_ulonglong, C.c_uint, C.c_uint, i, vp]
        L.ofk_feas_simulation.argtypes = [vp, vp, vp, vp, vp, i, vp, i, vp, vp, vp]
        L.ofk_hist_overlap.argtypes = [vp, vp, i, vp, i, i, C.POINTER(i)]
        L.ofk_pairs_upload.argtypes = [vp, vp, vp, i, i, i]
        L.ofk_pairs_upload_jpeg.argtypes = [vp, vp, vp, vp, vp, i]
        L.ofk_jpeg_stage.argtypes = [vp, i, vp, vp, i]; L.ofk_pairs_upload_staged.argtypes = [vp, i]
        L.ofk_jpeg_stage_error.restype = C.c_char_p; L.ofk_jpeg_stage_error.argtypes = [vp, i]
        L.ofk_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
        L.ofk_jpeg_destuff.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), i, C.POINTER(i)]
        L.ofk_jpeg_decode_bgr8.argtypes = [vp, vp, vp, i, vp]
        L.ofk_pairs_set_sensors.argtypes = [vp, vp, i]
        L.ofk_pairs_run.argtypes = [vp, C.POINTER(Params)]
        L.ofk_pairs_download.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.ofk_pairs_export_records_f32.argtypes = [vp, vp, i]
        L.ofk_stream_begin.argtypes = [vp, vp, i, i, i, C.POINTER(Params), vp, vp]
        L.ofk_stream_step.argtypes = [vp, vp, vp, C.POINTER(Params), i, i, vp, vp, vp]
        L.ofk_stream_begin_jpeg.argtypes = [vp, vp, vp, i, C.POINTER(Params), vp, vp]
        L.ofk_stream_step_jpeg.argtypes = [vp, vp, vp, vp, C.POINTER(Params), i, i, vp, vp, vp]
        L.ofk_imu_reset.argtypes = [vp, vp, i]; L.ofk_imu_push.argtypes = [vp, vp, vp, i, i]; L.ofk_imu_state.argtypes = [vp, vp, vp, i]
        L.ofk_filter_configure.argtypes = [vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, i]; L.ofk_filter_state.argtypes = [vp, vp, vp, i]
        L.ofk_stream_step_fused.argtypes = [vp, vp, vp, C.POINTER(Params), C.POINTER(Fusion), i, i, vp, vp, vp, vp]
        L.ofk_stream_step_fused_jpeg.argtypes = [vp, vp, vp, vp, C.POINTER(Params), C.POINTER(Fusion), i, i, vp, vp, vp, vp]
        L.ofk_stream_last_points.argtypes = [vp, vp, vp, i]
        L.ofk_pairs_filter_step.argtypes = [vp, d, i, i]
        L.ofk_set_streams.argtypes = [vp, i]
        L.ofk_set_overlap.argtypes = [vp, i]
        L.ofk_set_tuning.argtypes = [C.c_char_p, i]; L.ofk_get_tuning.argtypes = [C.c_char_p, C.POINTER(i)]
        L.ofk_mark.argtypes = [vp, i]; L.ofk_mark_wait.argtypes = [vp, i]
        L.ofk_profile_enable.argtypes = [vp, i]
        L.ofk_profile_read.argtypes = [vp, vp, vp]
        L.ofk_resident_pyramid.argtypes = [vp, i, i, vp, C.c_size_t]
        L.ofk_comm_unique_id.argtypes = [vp, i]; L.ofk_comm_init.argtypes = [vp, vp, i, i, i]; L.ofk_comm_add.argtypes = [vp, vp]; L.ofk_comm_destroy.argtypes = [vp]
        L.ofk_comm_rank.argtypes = [vp]; L.ofk_comm_world.argtypes = [vp]
        L.ofk_comm_gather_records.argtypes = [vp, i, i]; L.ofk_comm_fetch_records.argtypes = [vp, i, i, vp]
        L.ofk_comm_allreduce_f64.argtypes = [vp, vp, i, i]
        L.ofk_comm_count.argtypes = [vp]; L.ofk_comm_pending.argtypes = [vp, i]; L.ofk_comm_reorder_records.argtypes = [vp, i, i, i, vp]
        for s in SYMBOLS:
            if s not in ("ofk_last_error", "ofk_jpeg_stage_error"):      # the two that return a message, not a status
                getattr(L, s).restype = i
        _lib = L
        return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and a.shape != tuple(shape):
        a = np.ascontiguousarray(np.broadcast_to(a, shape))
    return a


def _opt(a, dtype, shape):
    return None if a is None else _arr(a, dtype, shape)


class Context:
    """One device + one HIP stream + all device buffers.  Calls are serialised by an internal lock
    (the C library is not thread-safe; rospy invokes callbacks from several threads)."""

    def __init__(self, device=0, max_w=1920, max_h=1080, max_batch=1, max_pts=512, max_level=3):
        self._L = load_library()
        self._h = C.c_void_p()
        self._lock = threading.RLock()
        rc = self._L.ofk_create(device, max_w, max_h, max_batch, max_pts, max_level, C.byref(self._h))
        if rc != OK:
            raise OfkError(rc, self._L.ofk_last_error(None).decode())
        self.device, self.max_w, self.max_h = device, max_w, max_h
        self.max_batch, self.max_pts, self.max_level = max_batch, max_pts, max_level
        self._resident = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.ofk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != OK:
            raise OfkError(rc, self._L.ofk_last_error(self._h).decode())

    def sync(self):
        with self._lock:
            self._ck(self._L.ofk_sync(self._h))

    # ------------------------------------------------------------------ image stages
    @staticmethod
    def _batched(img, nd):
        img = np.asarray(img)
        single = img.ndim == nd
        return (img[None] if single else img), single

    def gray_bgr8(self, bgr):
        bgr, single = self._batched(bgr, 3)
        bgr = _arr(bgr, np.uint8)
        B, h, w, ch = bgr.shape
        if ch != 3:
            raise ValueError("expected [..., h, w, 3] BGR")
        out = np.empty((B, h, w), np.uint8)
        with self._lock:
            self._ck(self._L.ofk_gray_bgr8(self._h, _p(bgr), B, h, w, _p(out)))
        return out[0] if single else out

    def pyr_down(self, src):
        src, single = self._batched(src, 2)
        src = _arr(src, np.uint8)
        B, h, w = src.shape
        out = np.empty((B, (h + 1) // 2, (w + 1) // 2), np.uint8)
        with self._lock:
            self._ck(self._L.ofk_pyr_down_u8(self._h, _p(src), B, h, w, _p(out)))
        return out[0] if single else out

    def pyramid(self, gray, max_level):
        """Levels 1..L of the pyramid calcOpticalFlowPyrLK builds: list of [B,h_l,w_l] uint8 arrays (a list of 2-D arrays for one image)."""
        gray, single = self._batched(gray, 2)
        gray = _arr(gray, np.uint8)
        B, h, w = gray.shape
        shapes, hh, ww = [], h, w
        for _ in range(int(max_level)):
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
            shapes.append((hh, ww))
        total = sum(a * b for a, b in shapes)
        out = np.empty((B, max(total, 1)), np.uint8)
        built = C.c_int()
        with self._lock:
            self._ck(self._L.ofk_pyramid_u8(self._h, _p(gray), B, h, w, int(max_level), _p(out), C.byref(built)))
        levels, o = [], 0
        stride = sum(a * b for a, b in shapes[:built.value])
        flat = out.reshape(-1)[:B * stride].reshape(B, stride) if stride else out[:, :0]
        for (a, b) in shapes[:built.value]:
            lvl = flat[:, o:o + a * b].reshape(B, a, b)
            levels.append(lvl[0] if single else lvl)
            o += a * b
        return levels

    def scharr(self, gray):
        gray, single = self._batched(gray, 2)
        gray = _arr(gray, np.uint8)
        B, h, w = gray.shape
        out = np.empty((B, h, w, 2), np.int16)
        with self._lock:
            self._ck(self._L.ofk_scharr_s16(self._h, _p(gray), B, h, w, _p(out)))
        return out[0] if single else out

    def mineig(self, gray, block_size):
        gray, single = self._batched(gray, 2)
        gray = _arr(gray, np.uint8)
        B, h, w = gray.shape
        out = np.empty((B, h, w), np.float32)
        with self._lock:
            self._ck(self._L.ofk_mineig_response(self._h, _p(gray), B, h, w, int(block_size), _p(out)))
        return out[0] if single else out

    def select_corners(self, eig, max_corners, quality, min_distance, mask=None):
        eig, single = self._batched(eig, 2)
        eig = _arr(eig, np.float32)
        B, h, w = eig.shape
        mask = _opt(mask, np.uint8, (B, h, w))
        pts = np.zeros((B, max_corners, 2), np.float32); cnt = np.zeros(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_select_corners(self._h, _p(eig), _p(mask), B, h, w, int(max_corners), float(quality),
                                                float(min_distance), _p(pts), _p(cnt)))
        return (pts[0], int(cnt[0])) if single else (pts, cnt)

    def good_features(self, gray, max_corners, quality, min_distance, block_size, mask=None):
        """Batched goodFeaturesToTrack.  Returns (pts [B,max_corners,2] f32, counts [B]) or, for a single
        image, the OpenCV-shaped (N,1,2) float32 array."""
        gray, single = self._batched(gray, 2)
        gray = _arr(gray, np.uint8)
        B, h, w = gray.shape
        mask = _opt(mask, np.uint8, (B, h, w))
        pts = np.zeros((B, max_corners, 2), np.float32); cnt = np.zeros(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_good_features(self._h, _p(gray), _p(mask), B, h, w, int(max_corners), float(quality),
                                               float(min_distance), int(block_size), _p(pts), _p(cnt)))
        if single:
            n = int(cnt[0])
            return pts[0, :n].reshape(n, 1, 2).copy()
        return pts, cnt

    def lk_pyr(self, prev, nxt, prev_pts, counts=None, win=15, max_level=3, max_count=20, eps=0.03, min_eig_thr=1e-4):
        """Batched calcOpticalFlowPyrLK.  Single image: prev_pts (N,1,2)/(N,2) -> (next (N,1,2), status (N,1), err (N,1)).
        Batch: prev_pts [B,S,2] + counts [B] -> (next [B,S,2], status [B,S], err [B,S])."""
        prev, single = self._batched(prev, 2)
        nxt, _ = self._batched(nxt, 2)
        prev = _arr(prev, np.uint8); nxt = _arr(nxt, np.uint8)
        if nxt.shape != prev.shape:
            raise ValueError(f"lk_pyr: next frame {nxt.shape} does not match the previous frame {prev.shape}")
        B, h, w = prev.shape
        if single:
            pp = _arr(prev_pts, np.float32).reshape(1, -1, 2)
            counts = np.array([pp.shape[1]], np.int32)
        else:
            pp = _arr(prev_pts, np.float32)
            counts = _arr(counts, np.int32, (B,))
        if pp.ndim != 3 or pp.shape[0] != B or pp.shape[2] != 2:
            raise ValueError(f"lk_pyr: prev_pts {pp.shape} is not [{B}, S, 2]")
        S = pp.shape[1]
        if np.any(counts < 0) or np.any(counts > S):
            raise ValueError("lk_pyr: counts outside 0..S")
        if S == 0:
            z = np.zeros((0, 1, 2), np.float32)
            return z, np.zeros((0, 1), np.uint8), np.zeros((0, 1), np.float32)
        nxt_pts = np.zeros((B, S, 2), np.float32); st = np.zeros((B, S), np.uint8); err = np.zeros((B, S), np.float32)
        with self._lock:
            self._ck(self._L.ofk_lk_pyr(self._h, _p(prev), _p(nxt), B, h, w, _p(pp), _p(counts), S, int(win), int(max_level),
                                        int(max_count), float(eps), float(min_eig_thr), _p(nxt_pts), _p(st), _p(err)))
        if single:
            return nxt_pts[0].reshape(S, 1, 2), st[0].reshape(S, 1), err[0].reshape(S, 1)
        return nxt_pts, st, err

    # ------------------------------------------------------------------ estimation
    def flow_model(self, x, v, omega, d, nrm, t=None):
        x = _arr(x, np.float64)
        if x.ndim not in (2, 3) or x.shape[-1] != 2:
            raise ValueError(f"flow_model: x {x.shape} must be [..., n, 2]")
        single = x.ndim == 2
        if single:
            x = x[None]
        B, n, _ = x.shape
        v = _arr(v, np.float64, (B, 3)); omega = _arr(omega, np.float64, (B, 3)); nrm = _arr(nrm, np.float64, (B, 3))
        d = _arr(d, np.float64, (B,)); t = _opt(t, np.float64, (B, 3))
        out = np.empty((B, n, 2), np.float64)
        if n:
            with self._lock:
                self._ck(self._L.ofk_flow_model(self._h, _p(x), B, n, _p(v), _p(omega), _p(d), _p(nrm), _p(t), _p(out)))
        return out[0] if single else out

    def feasibility(self, variant, x, u, nrm, v, dist=None, omega=None, t=None):
        x = _arr(x, np.float64); u = _arr(u, np.float64)
        if u.shape != x.shape or x.shape[-1] != 2:
            raise ValueError(f"feasibility: x {x.shape} and u {u.shape} must both be [..., n, 2]")
        single = x.ndim == 2
        if single:
            x = x[None]; u = u[None]
        B, n, _ = x.shape
        nrm = _arr(nrm, np.float64, (B, 3)); v = _arr(v, np.float64, (B, 3))
        dist = _opt(dist, np.float64, (B,)); omega = _opt(omega, np.float64, (B, 3)); t = _opt(t, np.float64, (B, 3))
        r = np.empty((B, n), np.float64); dd = np.empty((B, n), np.float64)
        if n:
            with self._lock:
                self._ck(self._L.ofk_feasibility(self._h, int(variant), _p(x), _p(u), B, n, _p(nrm), _p(v), _p(dist), _p(omega),
                                                 _p(t), _p(r), _p(dd)))
        return (r[0], dd[0]) if single else (r, dd)

    def velocity_solve(self, variant, x, u, d=None, nrm=None, omega=None, t=None, wgt=None, valid=None):
        """Returns out [B,8] = v[3], residual SS, rank, s[3] (single problem: [8])."""
        x = _arr(x, np.float64); u = _arr(u, np.float64)
        if u.shape[:-1] != x.shape[:-1] or x.shape[-1] != 2 or u.shape[-1] < 2:
            raise ValueError(f"velocity_solve: x {x.shape} must be [..., n, 2] and u {u.shape} [..., n, >=2] over the same points")
        single = x.ndim == 2
        if single:
            x = x[None]; u = u[None]
        B, n, _ = x.shape
        u = _arr(u[..., :2], np.float64)
        nrm = _arr(nrm, np.float64, (B, 3))
        d = _opt(d, np.float64, (B,)); omega = _opt(omega, np.float64, (B, 3)); t = _opt(t, np.float64, (B, 3))
        wgt = _opt(wgt, np.float64, (B, n)); valid = _opt(valid, np.uint8, (B, n))
        out = np.zeros((B, SOLVE_DOUBLES), np.float64)
        if n:
            with self._lock:
                self._ck(self._L.ofk_velocity_solve(self._h, int(variant), _p(x), _p(u), _p(valid), B, n, _p(d), _p(nrm),
                                                    _p(omega), _p(t), _p(wgt), _p(out)))
        return out[0] if single else out

    def imu_propagate(self, state, msg):
        state = _arr(state, np.float64).copy(); msg = _arr(msg, np.float64)
        single = state.ndim == 1
        s2 = state.reshape(-1, IMU_STATE); m2 = msg.reshape(-1, IMU_MSG)
        with self._lock:
            self._ck(self._L.ofk_imu_propagate(self._h, _p(s2), _p(m2), len(s2)))
        return s2[0] if single else s2

    def post_solve(self, v_obs, rotation, ang, offset):
        v_obs = _arr(v_obs, np.float64)
        single = v_obs.ndim == 1
        v2 = v_obs.reshape(-1, 3); B = len(v2)
        R = _arr(np.asarray(rotation, np.float64).reshape(-1, 9), np.float64, (B, 9))
        ang = _arr(ang, np.float64, (B, 3)); offset = _arr(offset, np.float64, (B, 3))
        out = np.empty((B, 3), np.float64)
        with self._lock:
            self._ck(self._L.ofk_post_solve(self._h, _p(v2), _p(R), _p(ang), _p(offset), B, _p(out)))
        return out[0] if single else out

    def feature_eval(self, pos, pos_err, oldpos, oldpos_err, vel, vel_err, focal_len, dummy_value, img_dim, weight, counts=None):
        """calc_height + dynamic_immobile + eval_ft of of_library.py for a batch of track sets ([batch, n, 2] positions;
        a single [n, 2] set is accepted).  Returns dict(height, height_err, immobile, score, order, bad_height)."""
        pos = _arr(pos, np.float64)
        single = pos.ndim == 2
        pos = pos.reshape((1,) + pos.shape) if single else pos
        B, n = pos.shape[0], pos.shape[1]
        def per_feature(a):                                      # scalar, [n] or [B, n]
            a = np.asarray(a, np.float64)
            return _arr(np.broadcast_to(a if a.size == 1 else a.reshape(-1, n), (B, n)), np.float64)
        pos_err = per_feature(pos_err)
        oldpos = _arr(np.asarray(oldpos, np.float64).reshape(B, n, 2), np.float64)
        oldpos_err = per_feature(oldpos_err)
        vel = _arr(np.broadcast_to(np.asarray(vel, np.float64).reshape(-1, 3), (B, 3)), np.float64)
        vel_err = _arr(np.broadcast_to(np.asarray(vel_err, np.float64).reshape(-1, 3), (B, 3)), np.float64)
        cn = np.full(B, n, np.int32) if counts is None else _arr(counts, np.int32, (B,))
        w = _arr(weight, np.float64, (4,))
        out = dict(height=np.zeros((B, n)), height_err=np.zeros((B, n)), immobile=np.zeros((B, n), np.uint8), score=np.zeros((B, n)),
                   order=np.zeros((B, n), np.int32))
        bad = C.c_int(0)
        with self._lock:
            self._ck(self._L.ofk_feature_eval(self._h, _p(pos), _p(pos_err), _p(oldpos), _p(oldpos_err), _p(cn), B, n, _p(vel), _p(vel_err),
                                              float(focal_len), float(dummy_value), int(img_dim[0]), int(img_dim[1]), _p(w),
                                              _p(out["height"]), _p(out["height_err"]), _p(out["immobile"]), _p(out["score"]),
                                              _p(out["order"]), C.byref(bad)))
        out["bad_height"] = bool(bad.value)
        if single:
            out = {k: (v[0] if isinstance(v, np.ndarray) else v) for k, v in out.items()}
        return out

    def d_split(self, d, d_exp_err, counts=None):
        """node:249-252: sorted plane distances, their consecutive differences and the number of gaps >= d_exp_err, per set
        ([batch, n] or a single [n])."""
        d = _arr(d, np.float64)
        single = d.ndim == 1
        d = d.reshape(1, -1) if single else d
        B, n = d.shape
        cn = np.full(B, n, np.int32) if counts is None else _arr(counts, np.int32, (B,))
        srt = np.zeros((B, n)); dif = np.zeros((B, n)); ns = np.zeros(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_d_split(self._h, _p(d), _p(cn), B, n, float(d_exp_err), _p(srt), _p(dif), _p(ns)))
        return (srt[0], dif[0, :max(n - 1, 0)], int(ns[0])) if single else (srt, dif, ns)

    def associate_sensors(self, t_img, imu_t, imu_quat, imu_omega, hgt_t, hgt_range, sensors=None):
        """evaluate_exp.py:68-95 on the device: nearest IMU / range sample per image time -> (sensors, imu_index, hgt_index).
        `sensors` ([n_img, 28]) supplies the fields the association does not touch (offset, scaling, centre, prior)."""
        t_img = _arr(np.atleast_1d(t_img), np.float64); n = len(t_img)
        imu_t = _arr(imu_t, np.float64); ni = len(imu_t)
        q = _arr(imu_quat, np.float64, (ni, 4)); w = _arr(imu_omega, np.float64, (ni, 3))
        hgt_t = _arr(hgt_t, np.float64); nh = len(hgt_t); r = _arr(hgt_range, np.float64, (nh,))
        out = make_sensors(n) if sensors is None else np.array(_arr(sensors, np.float64, (n, SENSOR_DOUBLES)))
        ii = np.empty(n, np.int32); hi = np.empty(n, np.int32)
        with self._lock:
            self._ck(self._L.ofk_associate_sensors(self._h, _p(t_img), n, _p(imu_t), _p(q), _p(w), ni, _p(hgt_t), _p(r), nh,
                                                   _p(out), _p(ii), _p(hi)))
        return out, ii, hi

    def kf_predict_update(self, F, H, Q, R, x, P, B=None, u=None, z=None, do_predict=True):
        F = _arr(F, np.float64); H = _arr(H, np.float64); Q = _arr(Q, np.float64); R = _arr(R, np.float64)
        ns = F.shape[0]; nm = H.shape[0]
        x = _arr(x, np.float64).copy(); P = _arr(P, np.float64).copy()
        single = x.ndim == 1
        x2 = x.reshape(-1, ns); batch = len(x2); P2 = P.reshape(batch, ns, ns)
        nc = 0
        if B is not None and u is not None:
            B = _arr(B, np.float64); nc = B.shape[1]; u = _arr(u, np.float64, (batch, nc))
        else:
            B = None; u = None
        z = _opt(z, np.float64, (batch, nm))
        with self._lock:
            self._ck(self._L.ofk_kf_predict_update(self._h, ns, nm, nc, _p(F), _p(B), _p(H), _p(Q), _p(R), _p(x2), _p(P2), _p(u),
                                                   _p(z), batch, 1 if do_predict else 0))
        return (x2[0], P2[0]) if single else (x2, P2)

    def of_simulation(self, truth, sig, pos, true_flow, z):
        truth = _arr(truth, np.float64, (13,)); sig = _arr(sig, np.float64, (6,))
        pos = _arr(pos, np.float64); true_flow = _arr(true_flow, np.float64)
        n = len(pos)
        z = _arr(z, np.float64).reshape(-1, 10 + 4 * n)
        trials = len(z)
        v = np.empty((trials, 3), np.float64); bound = np.empty(trials, np.float64)
        with self._lock:
            self._ck(self._L.ofk_of_simulation(self._h, _p(truth), _p(sig), _p(pos), _p(true_flow), n, _p(z), trials, _p(v), _p(bound)))
        return v, bound

    def of_simulation_rng(self, truth, sig, pos, true_flow, seed, step, trials, trial0=0):
        """of_simulation with the normals drawn on the device (ofk_of_simulation_rng: Philox4x32-10 + Box-Muller keyed by (seed, step,
        global trial index)).  Returns (v_obs [trials, 3], bound [trials])."""
        truth = _arr(truth, np.float64, (13,)); sig = _arr(sig, np.float64, (6,))
        pos = _arr(pos, np.float64); true_flow = _arr(true_flow, np.float64)
        n = len(pos)
        v = np.empty((int(trials), 3), np.float64); bound = np.empty(int(trials), np.float64)
        with self._lock:
            self._ck(self._L.ofk_of_simulation_rng(self._h, _p(truth), _p(sig), _p(pos), _p(true_flow), n, C.c_ulonglong(int(seed)), C.c_uint(int(step)),
                                                   C.c_uint(int(trial0)), int(trials), _p(v), _p(bound)))
        return v, bound

    def noise_normals(self, seed, step, trial, count):
        """Elements 0 .. count - 1 of the device generator's row for (seed, step, trial)."""
        out = np.empty(int(count), np.float64)
        with self._lock:
            self._ck(self._L.ofk_noise_normals(self._h, C.c_ulonglong(int(seed)), C.c_uint(int(step)), C.c_uint(int(trial)), int(count), _p(out)))
        return out

    def feas_simulation(self, truth, sig, pos, true_flow, z, per_trial=False):
        """simulation.py:70-104 for all trials in one launch.  Returns (mean [6, n] in the reference's return order, v_obs
        [trials, 3]) and, with per_trial=True, the per-trial table [trials, 6, n] as a third item."""
        truth = _arr(truth, np.float64, (16,)); sig = _arr(sig, np.float64, (7,))
        pos = _arr(pos, np.float64); true_flow = _arr(true_flow, np.float64)
        n = len(pos)
        if pos.shape != (n, 2) or true_flow.shape != (n, 2):
            raise ValueError("feas_simulation: pos and true_flow must be [n, 2]")
        z = _arr(z, np.float64).reshape(-1, 12 + 4 * n)
        trials = len(z)
        mean = np.empty((6, n), np.float64); v = np.empty((trials, 3), np.float64)
        per = np.empty((trials, 6, n), np.float64) if per_trial else None
        with self._lock:
            self._ck(self._L.ofk_feas_simulation(self._h, _p(truth), _p(sig), _p(pos), _p(true_flow), n, _p(z), trials, _p(mean), _p(per), _p(v)))
        return (mean, v, per) if per_trial else (mean, v)

    def hist_overlap(self, data1, data2, bins=100):
        """overlap(data1, data2) of simulation.py:124-136."""
        d1 = _arr(np.ravel(data1), np.float64); d2 = _arr(np.ravel(data2), np.float64)
        out = C.c_int(0)
        with self._lock:
            self._ck(self._L.ofk_hist_overlap(self._h, _p(d1), len(d1), _p(d2), len(d2), int(bins), C.byref(out)))
        return int(out.value)

    # ------------------------------------------------------------------ resident frame-pair pipeline
    def pairs_upload(self, prev_bgr, next_bgr):
        prev_bgr, _ = self._batched(prev_bgr, 3); next_bgr, _ = self._batched(next_bgr, 3)
        prev_bgr = _arr(prev_bgr, np.uint8); next_bgr = _arr(next_bgr, np.uint8)
        B, h, w, ch = prev_bgr.shape
        if ch != 3 or next_bgr.shape != prev_bgr.shape:
            raise ValueError("expected two [B,h,w,3] BGR batches of equal shape")
        with self._lock:
            self._ck(self._L.ofk_pairs_upload(self._h, _p(prev_bgr), _p(next_bgr), B, h, w))
        self._resident = (B, h, w)

    @staticmethod
    def _jpeg_args(streams):
        bufs = [bytes(s) for s in streams]
        ptrs = (C.c_char_p * len(bufs))(*bufs)
        sizes = (C.c_size_t * len(bufs))(*[len(b) for b in bufs])
        return bufs, ptrs, sizes

    def pairs_upload_jpeg(self, prev_streams, next_streams):
        """Resident frame pairs from baseline JPEG streams (CompressedImage payloads), decoded on the device."""
        if len(prev_streams) != len(next_streams) or not len(prev_streams):
            raise ValueError("expected two equally long, non-empty lists of JPEG streams")
        h, w, _ = jpeg_info(prev_streams[0])
        keep0, p0, s0 = self._jpeg_args(prev_streams)
        keep1, p1, s1 = self._jpeg_args(next_streams)
        with self._lock:
            self._ck(self._L.ofk_pairs_upload_jpeg(self._h, p0, s0, p1, s1, len(keep0)))
        self._resident = (len(keep0), h, w)

    def jpeg_stage(self, slot, streams):
        """Phase 1 of the compressed ingest (ofk_jpeg_stage): parse + staging copy + asynchronous H2D of `streams` into staging slot
        0 / 1.  Takes no context lock: it is meant to run on a helper thread while the owner thread decodes the other slot.
        Returns (count, h, w)."""
        if not len(streams):
            raise ValueError("no streams")
        h, w, _ = jpeg_info(streams[0])
        keep, ptrs, sizes = self._jpeg_args(streams)
        rc = self._L.ofk_jpeg_stage(self._h, int(slot), ptrs, sizes, len(keep))
        if rc != OK:                                             # the slot's own message: the owner thread may be writing the context's
            raise OfkError(rc, self._L.ofk_jpeg_stage_error(self._h, int(slot)).decode())
        return len(keep), h, w

    def pairs_upload_staged(self, slot, staged):
        """Phase 2 (ofk_pairs_upload_staged): decodes the 2 B streams staged in `slot` (B previous frames, then B next frames) into
        the resident pair buffers.  `staged` = what jpeg_stage returned for that slot."""
        count, h, w = staged
        with self._lock:
            self._ck(self._L.ofk_pairs_upload_staged(self._h, int(slot)))
        self._resident = (count // 2, h, w)

    def jpeg_decode(self, streams):
        """[B,h,w,3] BGR uint8 from B baseline JPEG streams of one size and sampling (cv2.imdecode, batched)."""
        if not len(streams):
            raise ValueError("no streams")
        h, w, _ = jpeg_info(streams[0])
        keep, ptrs, sizes = self._jpeg_args(streams)
        out = np.empty((len(keep), h, w, 3), np.uint8)
        with self._lock:
            self._ck(self._L.ofk_jpeg_decode_bgr8(self._h, ptrs, sizes, len(keep), _p(out)))
        return out

    def pairs_set_sensors(self, sensors):
        s = _arr(sensors, np.float64).reshape(-1, SENSOR_DOUBLES)
        with self._lock:
            self._ck(self._L.ofk_pairs_set_sensors(self._h, _p(s), len(s)))

    def pairs_run(self, params):
        with self._lock:
            self._ck(self._L.ofk_pairs_run(self._h, C.byref(params)))

    def pairs_download(self, points=True):
        B = self._resident[0]; S = self.max_pts
        rec = np.empty((B, RECORD_DOUBLES), np.float64)
        if points:
            pp = np.empty((B, S, 2), np.float32); npts = np.empty((B, S, 2), np.float32)
            st = np.empty((B, S), np.uint8); err = np.empty((B, S), np.float32)
        else:
            pp = npts = st = err = None
        cnt = np.empty(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_pairs_download(self._h, _p(rec), _p(pp), _p(npts), _p(st), _p(err), _p(cnt)))
        return dict(records=rec, prev_pts=pp, next_pts=npts, status=st, err=err, counts=cnt)

    def pairs_export_records_f32(self, device_ptr, batch):
        with self._lock:
            self._ck(self._L.ofk_pairs_export_records_f32(self._h, C.c_void_p(int(device_ptr)), int(batch)))

    # ------------------------------------------------------------------ video streams (persistent tracks on the device)
    def stream_begin(self, first_bgr, params):
        first_bgr, _ = self._batched(first_bgr, 3)
        first_bgr = _arr(first_bgr, np.uint8)
        B, h, w, ch = first_bgr.shape
        if ch != 3:
            raise ValueError("expected [B,h,w,3] BGR frames")
        mc = int(params.max_corners)
        tracks = np.zeros((B, mc, 2), np.float32); counts = np.zeros(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_stream_begin(self._h, _p(first_bgr), B, h, w, C.byref(params), _p(tracks), _p(counts)))
        self._stream = (B, h, w)
        return tracks, counts

    def stream_step(self, next_bgr, sensors, params, min_features, mask_radius):
        B, h, w = self._stream
        next_bgr, _ = self._batched(next_bgr, 3)
        next_bgr = _arr(next_bgr, np.uint8, (B, h, w, 3))
        sensors = _arr(sensors, np.float64).reshape(B, SENSOR_DOUBLES)
        mc = int(params.max_corners)
        rec = np.zeros((B, RECORD_DOUBLES), np.float64)
        tracks = np.zeros((B, mc, 2), np.float32); counts = np.zeros(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_stream_step(self._h, _p(next_bgr), _p(sensors), C.byref(params), int(min_features), int(mask_radius),
                                             _p(rec), _p(tracks), _p(counts)))
        return rec, tracks, counts

    def stream_begin_jpeg(self, streams, params):
        """stream_begin with one baseline JPEG stream per camera (CompressedImage payloads), decoded on the device."""
        h, w, _ = jpeg_info(streams[0])
        keep, ptrs, sizes = self._jpeg_args(streams)
        B, mc = len(keep), int(params.max_corners)
        tracks = np.zeros((B, mc, 2), np.float32); counts = np.zeros(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_stream_begin_jpeg(self._h, ptrs, sizes, B, C.byref(params), _p(tracks), _p(counts)))
        self._stream = (B, h, w)
        return tracks, counts

    def stream_step_jpeg(self, streams, sensors, params, min_features, mask_radius):
        B, h, w = self._stream
        if len(streams) != B:
            raise ValueError(f"expected {B} JPEG streams")
        keep, ptrs, sizes = self._jpeg_args(streams)
        sensors = _arr(sensors, np.float64).reshape(B, SENSOR_DOUBLES)
        mc = int(params.max_corners)
        rec = np.zeros((B, RECORD_DOUBLES), np.float64)
        tracks = np.zeros((B, mc, 2), np.float32); counts = np.zeros(B, np.int32)
        with self._lock:
            self._ck(self._L.ofk_stream_step_jpeg(self._h, ptrs, sizes, _p(sensors), C.byref(params), int(min_features), int(mask_radius),
                                                  _p(rec), _p(tracks), _p(counts)))
        return rec, tracks, counts

    # ------------------------------------------------------------------ resident per-stream filters + fused step
    def imu_reset(self, batch, state0=None):
        st = None if state0 is None else _arr(state0, np.float64, (IMU_STATE,))
        with self._lock:
            self._ck(self._L.ofk_imu_reset(self._h, _p(st), int(batch)))

    def imu_push(self, msgs, counts=None):
        """msgs [B, M, 15] (or [B, 15] for one message per stream); counts [B] (default: M each)."""
        m = _arr(msgs, np.float64)
        if m.ndim == 2:
            m = m[:, None, :]
        B, M, k = m.shape
        if k != IMU_MSG:
            raise ValueError(f"an IMU message has {IMU_MSG} fields")
        cn = np.full(B, M, np.int32) if counts is None else _arr(counts, np.int32, (B,))
        m = np.ascontiguousarray(m)
        with self._lock:
            self._ck(self._L.ofk_imu_push(self._h, _p(m), _p(cn), M, B))

    def imu_state(self, batch):
        st = np.empty((batch, IMU_STATE), np.float64); dv = np.empty((batch, 3), np.float64)
        with self._lock:
            self._ck(self._L.ofk_imu_state(self._h, _p(st), _p(dv), int(batch)))
        return st, dv

    def filter_configure(self, model, batch):
        """model: pipeline.FilterModel (F, B, H, Q, R, x0, P0)."""
        ns, nm, nc = model.ns, model.nm, model.nc
        F = _arr(model.F, np.float64, (ns, ns)); H = _arr(model.H, np.float64, (nm, ns)); Q = _arr(model.Q, np.float64, (ns, ns))
        R = _arr(model.R, np.float64, (nm, nm)); Bm = _arr(model.B, np.float64, (ns, nc)) if nc else None
        x0 = _arr(model.x0, np.float64, (ns,)); P0 = _arr(model.P0, np.float64, (ns, ns))
        with self._lock:
            self._ck(self._L.ofk_filter_configure(self._h, ns, nm, nc, _p(F), _p(Bm), _p(H), _p(Q), _p(R), _p(x0), _p(P0), int(batch)))
        self._kf_ns = ns

    def filter_state(self, batch):
        ns = self._kf_ns
        x = np.empty((batch, ns), np.float64); P = np.empty((batch, ns, ns), np.float64)
        with self._lock:
            self._ck(self._L.ofk_filter_state(self._h, _p(x), _p(P), int(batch)))
        return x, P

    def stream_step_fused(self, next_bgr, sensors, params, fusion, min_features, mask_radius):
        """ofk_stream_step_fused: returns (records [B,16], fused [B,8], tracks [B,mc,2], counts [B])."""
        B, h, w = self._stream
        sensors = _arr(sensors, np.float64).reshape(B, SENSOR_DOUBLES)
        mc = int(params.max_corners)
        rec = np.zeros((B, RECORD_DOUBLES), np.float64); fused = np.zeros((B, 8), np.float64)
        tracks = np.zeros((B, mc, 2), np.float32); counts = np.zeros(B, np.int32)
        if isinstance(next_bgr, (list, tuple)) and isinstance(next_bgr[0], (bytes, bytearray, memoryview)):
            if len(next_bgr) != B:
                raise ValueError(f"expected {B} JPEG streams")
            keep, ptrs, sizes = self._jpeg_args(next_bgr)
            with self._lock:
                self._ck(self._L.ofk_stream_step_fused_jpeg(self._h, ptrs, sizes, _p(sensors), C.byref(params), C.byref(fusion), int(min_features),
                                                            int(mask_radius), _p(rec), _p(fused), _p(tracks), _p(counts)))
        else:
            frames, _ = self._batched(next_bgr, 3)
            frames = _arr(frames, np.uint8, (B, h, w, 3))
            with self._lock:
                self._ck(self._L.ofk_stream_step_fused(self._h, _p(frames), _p(sensors), C.byref(params), C.byref(fusion), int(min_features),
                                                       int(mask_radius), _p(rec), _p(fused), _p(tracks), _p(counts)))
        return rec, fused, tracks, counts

    def pairs_filter_step(self, batch, z_sign=-1.0, z_source=0):
        """Per-pair resident filter update behind the latest pairs_run (asynchronous)."""
        with self._lock:
            self._ck(self._L.ofk_pairs_filter_step(self._h, float(z_sign), int(z_source), int(batch)))

    def stream_last_points(self, stride):
        """(next_pts [B,stride,2] f32, keep [B,stride] u8) of the latest stream step."""
        B = self._stream[0]
        nxt = np.zeros((B, stride, 2), np.float32); keep = np.zeros((B, stride), np.uint8)
        with self._lock:
            self._ck(self._L.ofk_stream_last_points(self._h, _p(nxt), _p(keep), int(stride)))
        return nxt, keep

    # ------------------------------------------------------------------ multi-GPU exchange (RCCL through the library, no torch)
    def comm_init(self, unique_id, rank, world):
        uid = np.frombuffer(bytes(unique_id), np.uint8).copy()
        if uid.size == 0 or uid.size % 128:
            raise ValueError("RCCL unique ids have 128 bytes each")
        with self._lock:
            self._ck(self._L.ofk_comm_init(self._h, _p(uid), uid.size // 128, int(rank), int(world)))
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_add(self, unique_id):
        """One more communicator (128-byte id).  Collective: every rank calls it, in the same order; an error is fatal for the job."""
        uid = np.frombuffer(bytes(unique_id), np.uint8).copy()
        if uid.size != 128:
            raise ValueError("one 128-byte RCCL unique id")
        with self._lock:
            self._ck(self._L.ofk_comm_add(self._h, _p(uid)))

    def comm_destroy(self):
        with self._lock:
            self._ck(self._L.ofk_comm_destroy(self._h))

    def comm_gather_records(self, batch, slot=0):
        """Queues the all-gather of the latest step's [batch, 8] f32 records behind that step; returns at once."""
        with self._lock:
            self._ck(self._L.ofk_comm_gather_records(self._h, int(batch), int(slot)))

    def comm_fetch_records(self, batch, slot=0):
        out = np.empty((self.comm_world, int(batch), 8), np.float32)
        with self._lock:
            self._ck(self._L.ofk_comm_fetch_records(self._h, int(slot), int(batch), _p(out)))
        return out

    def comm_count(self):
        """Communicators the ranks agreed on (1 = one gather per step behind the last slice)."""
        return int(self._L.ofk_comm_count(self._h))

    def comm_pending(self, slot=0):
        """Non-blocking: bitmask of the slices whose gather of `slot` has not completed (watchdogs)."""
        return int(self._L.ofk_comm_pending(self._h, int(slot)))

    def comm_allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(np.atleast_1d(values), np.float64).copy()
        with self._lock:
            self._ck(self._L.ofk_comm_allreduce_f64(self._h, _p(v), v.size, {"sum": 0, "max": 1, "min": 2}[op]))
        return v

    def set_overlap(self, on):
        self._ck(self._L.ofk_set_overlap(self._h, 1 if on else 0))

    def mark(self, slot=0):
        """Records a completion mark behind everything queued on the context's stream (slots 0..7)."""
        self._ck(self._L.ofk_mark(self._h, int(slot)))

    def mark_wait(self, slot=0):
        """Blocks the host until mark `slot` has been reached; later work keeps running."""
        self._ck(self._L.ofk_mark_wait(self._h, int(slot)))

    def set_streams(self, n):
        self._ck(self._L.ofk_set_streams(self._h, int(n)))

    def profile_enable(self, mask):
        self._ck(self._L.ofk_profile_enable(self._h, int(mask)))

    def resident_pyramid(self, frame_set, image, h, w, levels):
        """The resident pyramid of one image (frame set 0 = previous, 1 = next frames of the latest pairs batch, or the stream
        loop's slot): list of uint8 arrays, level 0 = the gray frame."""
        shapes, offs, off = [], [], 0
        for l in range(levels + 1):
            shapes.append((h, w)); offs.append(off)
            off += (h * w + 255) // 256 * 256
            h, w = (h + 1) // 2, (w + 1) // 2
        buf = np.empty(off, np.uint8)
        with self._lock:
            self._ck(self._L.ofk_resident_pyramid(self._h, int(frame_set), int(image), _p(buf), buf.size))
        return [buf[o:o + a * b].reshape(a, b).copy() for (a, b), o in zip(shapes, offs)]

    def profile_read(self):
        ms = np.zeros(len(STAGES), np.float64); n = np.zeros(len(STAGES), np.int32)
        with self._lock:
            self._ck(self._L.ofk_profile_read(self._h, _p(ms), _p(n)))
        return {s: (float(ms[i]), int(n[i])) for i, s in enumerate(STAGES)}


def comm_unique_id(n_ids=1):
    """n_ids x 128-byte RCCL unique ids (call on rank 0, hand to every rank): one communicator per free-running slice."""
    uid = np.zeros(128 * int(n_ids), np.uint8)
    rc = load_library().ofk_comm_unique_id(_p(uid), int(n_ids))
    if rc != OK:
        raise OfkError(rc, load_library().ofk_last_error(None).decode())
    return uid.tobytes()


def comm_reorder_records(recv, world, batch, slices):
    """Host-only: the per-slice receive order of ofk_comm_gather_records -> rank-major [world, batch, 8] (ofk_comm_fetch_records'
    reassembly, callable without a GPU)."""
    recv = np.ascontiguousarray(recv, np.float32)
    out = np.empty((int(world), int(batch), 8), np.float32)
    L = load_library()
    if recv.size != out.size or L.ofk_comm_reorder_records(_p(recv), int(world), int(batch), int(slices), _p(out)) != OK:
        raise OfkError(E_INVALID, "comm_reorder_records: bad argument")
    return out


def set_tuning(knob, value):
    """Launch-geometry knob for measurements (ofk_set_tuning in include/ofk.h); 0 restores the built-in choice.  Process-wide."""
    L = load_library()
    if L.ofk_set_tuning(str(knob).encode(), int(value)) != OK:
        raise OfkError(E_INVALID, L.ofk_last_error(None).decode())


def get_tuning(knob):
    L = load_library()
    v = C.c_int(0)
    if L.ofk_get_tuning(str(knob).encode(), C.byref(v)) != OK:
        raise OfkError(E_INVALID, L.ofk_last_error(None).decode())
    return v.value


def make_sensors(batch, d=1.0, normal=(0, 0, 1), omega=(0, 0, 0), rotation=None, offset=(0, 0, 0.1), scaling=0.01,
                 cx=0.0, cy=0.0, v_prior=(0, 0, 0)):
    """Builds the [batch][28] sensor records ofk_pairs_set_sensors takes (layout: include/ofk.h)."""
    s = np.zeros((batch, SENSOR_DOUBLES), np.float64)
    s[:, 0] = d; s[:, 1:4] = normal; s[:, 4:7] = omega
    s[:, 7:16] = np.eye(3).ravel() if rotation is None else np.asarray(rotation, np.float64).reshape(-1, 9)
    s[:, 16:19] = offset; s[:, 19] = scaling; s[:, 20] = cx; s[:, 21] = cy; s[:, 22:25] = v_prior
    return s


_default = None
_default_lock = threading.Lock()


def default_context(min_w=0, min_h=0, min_pts=0, min_level=0):
    """Process-wide context used by the of_library / cv2-style facade; grown on demand."""
    global _default
    with _default_lock:
        c = _default
        if c is None or c.max_w * c.max_h < min_w * min_h or c.max_pts < min_pts or c.max_level < min_level:
            w = max(min_w, c.max_w if c else 1920); h = max(min_h, c.max_h if c else 1080)
            pts = max(min_pts, c.max_pts if c else 512); lvl = max(min_level, c.max_level if c else 5)
            # build the bigger context FIRST: if that fails (a corrupt JPEG header asking for 65535 x 65535, out of memory) the
            # old one stays in place and the facade keeps working
            new = Context(int(os.environ.get("OFK_DEVICE", "0")), w, h, 1, pts, lvl)
            _default = new
            if c is not None:
                c.close()
        return _default
